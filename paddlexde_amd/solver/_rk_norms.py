"""Norm reductions of the adaptive stepper: the scaled norms of the initial-step heuristic (paddlexde/solver/base_adaptive_solver.py:33-72)
and the plumbing every per-attempt norm shares — chunks of XDE_MAX_SEG segments for tuple states with many members, a user's norm
callable as framework ops, and the exchange of the per-segment sums across a batch-sharding process group (the one
coupling between the ranks).  A mixin of ``AdaptiveRKSolver`` (it reads the stepper's buffers and options)."""
import torch

from .. import _hip


class NormReductions:
    def _allreduce_sums(self, sums):
        if self.process_group is None:
            return
        if self.norm_exchange is not None and sums.is_cuda:
            self.norm_exchange.exchange(sums, self._norm_kind)
            return
        import torch.distributed as dist

        group = None if self.process_group is True else self.process_group
        buf = sums
        staged = sums.is_cuda and dist.get_backend(group) == "gloo"
        if staged:  # rehearsal transport (several ranks on one GPU): gloo reduces on the host
            buf = sums.cpu()
        if self._norm_kind == _hip.NORM_RMS:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(buf[: _hip.XDE_MAX_SEG], op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(buf[_hip.XDE_MAX_SEG :], op=dist.ReduceOp.SUM, group=group)
        if staged:
            sums.copy_(buf)

    def _global_counts(self):
        counts = list(self._seg_count_local)
        if self.process_group is not None:
            import torch.distributed as dist

            group = None if self.process_group is True else self.process_group
            cdev = "cpu" if dist.get_backend(group) == "gloo" else self.y0.device
            c = torch.tensor(counts, dtype=torch.float64, device=cdev)
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            counts = c.tolist()
        return counts

    def _user_norm(self, flat):
        """Apply a user-supplied norm callable to a flat state-like tensor (tuple state: to the tuple of views)."""
        if self._seg_shapes is not None:
            arg = tuple(flat[s : s + l].view(shape) for (s, l), shape in zip(self._segs, self._seg_shapes))
        else:
            arg = flat.view(self.y0.shape)
        v = self.norm(arg)
        v = v if torch.is_tensor(v) else torch.as_tensor(float(v))
        return v.detach().abs().to(device=flat.device, dtype=torch.float64).reshape(())

    def _scaled_norms(self, pairs, y0, rtol, atol):
        """norm(a / scale) or norm((a - b) / scale) for each (a, b) pair; one host read for all of them."""
        if self._custom_norm:
            scale = float(atol) + y0.abs() * float(rtol)
            vals = [self._user_norm(((a - b) if b is not None else a) / scale) for a, b in pairs]
            return torch.stack(vals).tolist()
        res = torch.empty(len(pairs), dtype=torch.float64, device=y0.device)
        for i, (a, b) in enumerate(pairs):
            self._scaled_norm_into(a, b, y0, rtol, atol, res[i : i + 1])
        return res.tolist()

    def _reduce_chunks(self, launch_partial, out, nonfinite_out=None):
        """Run one norm over all segments: ``launch_partial(xsegs)`` enqueues the partial kernel for one chunk of segments;
        the scalar norm (max over every segment) lands in ``out`` (device double[1])."""
        be = self.backend
        sdt = _hip.dtype_code(self.y0.dtype)
        m = _hip.XDE_MAX_SEG
        if self._chunks is None:
            launch_partial(self._xsegs)
            be.norm_finalize(self._ws, 0, self._sums)
            self._allreduce_sums(self._sums)
            be.norm_result(self._sums, self._seg_count, self._norm_kind, sdt, out)
            return
        cres = torch.empty(len(self._chunks), dtype=torch.float64, device=self.y0.device)
        nf = None
        for ci, (first, xs) in enumerate(self._chunks):
            launch_partial(xs)
            be.norm_finalize(self._ws, 0, self._sums)
            self._allreduce_sums(self._sums)
            be.norm_result(self._sums, self._seg_count[first : first + xs.n_seg], self._norm_kind, sdt, cres[ci : ci + 1])
            if nonfinite_out is not None:
                part = self._sums[m : m + xs.n_seg].sum()
                nf = part if nf is None else nf + part
        out.copy_(cres.max().reshape(out.shape))  # torch.max propagates NaN, like the kernels' max over segments
        if nonfinite_out is not None:
            nonfinite_out.copy_(nf.reshape(nonfinite_out.shape))

    def _scaled_norm_into(self, a, b, y0, rtol, atol, out):
        be = self.backend
        self._reduce_chunks(
            lambda xs: be.scaled_norm_partial(a, b, y0, float(rtol), float(atol), xs, self._norm_kind, self._ws, 0), out)
