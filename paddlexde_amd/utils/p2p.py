"""One-shot peer-to-peer exchange of the error-norm sums between the GPUs of one node (libxde_hip.so: xde_p2p_*).

``PeerExchange(group)`` gives every rank a mailbox in uncached device memory, ships the 64-byte IPC handles around with
``torch.distributed.all_gather_object`` (set-up only) and maps the peers' mailboxes; ``exchange(sums, norm_kind)`` then
stands in for ``all_reduce(sums)`` in the batch-sharded solve: ONE kernel launch per attempted step, xGMI stores straight
into the peers' memory, the sum taken in rank order on every GPU — no collective library and no host on the step's path, so
it also works under the speculative ("lag") pipeline.  Pass it to a solver as ``options={"process_group": ...,
"norm_exchange": PeerExchange(...)}``; ``torch.distributed`` stays the transport of everything that happens once per solve.
"""
import ctypes as C

import torch

from .. import _hip


class PeerExchange:
    capturable = True  # the exchange is one kernel launch whose counter lives in device memory: valid inside a replayed hipGraph
    # the solver takes finalize -> exchange -> controller of a sharded attempt as ONE launch (xde_p2p_rk_control) when this is set;
    # False keeps the three launches (xde_norm_finalize, xde_p2p_exchange, xde_rk_control) — same bits, measured side by side
    fused_control = True
    # Polls (~60 ns each) before an exchange gives up: about 20 s.  The wait has to cover the honest skew between ranks — one rank
    # still tuning its GEMMs or paging a library in while another has already enqueued its first exchange — and still end: a peer
    # that died or fell out of lock-step must not leave a wave spinning for ever.  (Tests that exercise the failure path set it to
    # milliseconds; bench.py's transport probe to a tenth of a second.)
    SPIN_LIMIT = 400_000_000

    def __init__(self, group=None, device=None):
        import torch.distributed as dist

        from .exchange import raise_together

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = None
        self._local = None
        self._opened = []
        # -- rank-local: library, mailbox, IPC handle.  Nothing collective has been entered when one of these fails, and the
        #    ranks agree on the outcome before the handles travel (a failure is an exception on EVERY rank).
        err, handle = None, None
        try:
            if self.world > _hip.XDE_P2P_MAX_RANKS:
                raise _hip.XdeError("PeerExchange serves one node: at most {} ranks".format(_hip.XDE_P2P_MAX_RANKS))
            self.lib = _hip.load_library()
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            with torch.cuda.device(self.device):
                p = C.c_void_p()
                self._check(self.lib.xde_p2p_alloc(C.byref(p)), "xde_p2p_alloc")
                self._local = p.value
                handle = (C.c_ubyte * _hip.XDE_P2P_HANDLE_BYTES)()
                self._check(self.lib.xde_p2p_export(self._local, handle), "xde_p2p_export")
        except Exception as e:  # noqa: BLE001
            err = e
        try:
            raise_together(err, "PeerExchange: mailbox set-up", group)
        except Exception:
            self._release_local()
            raise
        # -- collective: every rank's (device index, handle)
        infos = [None] * self.world
        dist.all_gather_object(infos, (self.device.index, bytes(handle)), group=group)
        self.peer_devices = [d for d, _ in infos]
        # -- rank-local: peer access, mapping the peers' mailboxes
        err = None
        try:
            with torch.cuda.device(self.device):
                ptrs = []
                for r, (d, h) in enumerate(infos):
                    if r == self.rank:
                        ptrs.append(self._local)
                        continue
                    if d != self.device.index and not torch.cuda.can_device_access_peer(self.device.index, d):
                        raise _hip.XdeError("device {} cannot access its peer device {} (rank {})".format(self.device.index, d, r))
                    q = C.c_void_p()
                    buf = (C.c_ubyte * _hip.XDE_P2P_HANDLE_BYTES).from_buffer_copy(h)
                    self._check(self.lib.xde_p2p_import(buf, C.byref(q)), "xde_p2p_import")
                    self._opened.append(q.value)
                    ptrs.append(q.value)
                self._peers = (C.c_void_p * self.world)(*ptrs)
        except Exception as e:  # noqa: BLE001
            err = e
        try:
            raise_together(err, "PeerExchange: mapping the peers' mailboxes", group)  # (also the barrier: every mailbox is mapped
        except Exception:                                                             #  everywhere before the first store into it)
            self._release_local()
            raise

    def abandon(self):
        """Rank-local release, no collective: for a group that is being torn down after a failure."""
        self._release_local()

    def _release_local(self):
        """Give back what THIS rank holds (no collective): the imported mappings, then the own mailbox."""
        try:
            for q in self._opened:
                self.lib.xde_p2p_close(q)
            self._opened = []
            if self._local is not None:
                self.lib.xde_p2p_free(self._local)
                self._local = None
        except Exception:  # noqa: BLE001
            pass

    def __del__(self):  # best effort for a forgotten close(): release what is ours, touch nothing a peer may still use
        try:
            if getattr(self, "_local", None) is not None:
                for q in self._opened:
                    self.lib.xde_p2p_close(q)
                self.lib.xde_p2p_free(self._local)
                self._local = None
        except Exception:
            pass

    def _check(self, rc, who):
        if rc != _hip.XDE_OK:
            raise _hip.XdeError("{} failed (status {}): {}".format(who, rc, self.lib.xde_last_error().decode()))

    def exchange(self, sums, norm_kind):
        """In place: ``sums`` (2*XDE_MAX_SEG device doubles) becomes the rank-ordered sum (max for a linf norm's values) over
        all ranks.  Enqueued on torch's current stream."""
        if not sums.is_cuda or sums.dtype != torch.float64 or sums.numel() != 2 * _hip.XDE_MAX_SEG:
            raise _hip.XdeError("PeerExchange.exchange takes the solver's 2*XDE_MAX_SEG float64 device sums")
        rc = self.lib.xde_p2p_exchange(sums.data_ptr(), self._local, self._peers, self.world, self.rank, int(norm_kind),
                                       self.SPIN_LIMIT, _hip.HipBackend._stream(sums))
        self._check(rc, "xde_p2p_exchange")

    def error_info(self):
        """``(exchange, reported_by)``: the number of the first failed exchange on this rank (0: none) and the rank whose wait ran
        out first when this rank was told by a peer (None: this rank's own wait ran out).  One blocking 24-byte read."""
        e = (C.c_int64 * 3)()
        self._check(self.lib.xde_p2p_error_info(self._local, e, 3, _hip.HipBackend._stream(torch.empty(0, device=self.device))),
                    "xde_p2p_error_info")
        return int(e[0]), (int(e[2]) - 1 if e[1] != 0 and e[2] > 0 else None)

    def error(self):
        """Exchange number of the first failed exchange on this rank, or 0."""
        return self.error_info()[0]

    def close(self):
        import torch.distributed as dist

        if self._local is None:
            return
        torch.cuda.synchronize(self.device)
        try:
            dist.barrier(group=self.group)  # nobody unmaps a mailbox a peer may still be storing into
        except Exception:
            pass
        for q in self._opened:
            self.lib.xde_p2p_close(q)
        self._opened = []
        self.lib.xde_p2p_free(self._local)
        self._local = None
