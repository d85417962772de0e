"""TEST DOUBLE of paddlexde_amd._hip.HipBackend on CPU tensors (numpy arithmetic).

Lives in tests/ on purpose: it lets `-m "not gpu"` tests drive the product's HOST logic (stepping loops,
operand plans, buffer rotation, output bookkeeping, tuple flattening, the adjoint, the process-group
reduction hook) in a container without a GPU.  It is never importable from the product package and the
product has no code path that selects it.  Each method states the kernel contract of include/xde_hip.h
in numpy, in the same op order as csrc/xde_*.hip.
"""
import ctypes as C

import numpy as np
import torch

from paddlexde_amd import _hip
from paddlexde_amd._hip import XdeCtrl

_NP = {torch.float32: np.float32, torch.float64: np.float64}


def _np(t):
    return t.detach().numpy()


class NumpyDoubleBackend:
    name = "numpy-double(test)"

    def __init__(self):
        self._slots = {}
        self.launches = []

    def require_device(self, *tensors):
        pass

    # -- allocation ---------------------------------------------------------------------------
    def new_ctrl(self, device):
        return torch.zeros(C.sizeof(XdeCtrl), dtype=torch.uint8)

    def new_workspace(self, device):
        return torch.zeros(64, dtype=torch.uint8)

    def new_sums(self, device):
        return torch.zeros(2 * _hip.XDE_MAX_SEG, dtype=torch.float64)

    def acquire_work(self, device, state_dtype):
        w = _hip._Work()
        w.key = None
        w.ctrl, w.ws, w.sums = self.new_ctrl(device), self.new_workspace(device), self.new_sums(device)
        w.t_stage = torch.zeros(_hip.XDE_MAX_STAGE, dtype=state_dtype)
        return w

    def release_work(self, w):
        pass

    @staticmethod
    def _c(ctrl) -> XdeCtrl:
        return XdeCtrl.from_buffer(ctrl.numpy())

    # -- K1 -------------------------------------------------------------------------------------
    def stage_combine(self, out, y0, ks, coef, mode, *, scale=1.0, dt_host=0.0, ctrl=None, y0_alt=None, k0_alt=None,
                      out2=None, coef2=None, damping=0.0, nt_mask=0):
        self.launches.append("combine")
        T = _NP[out.dtype]
        sel = 0
        if ctrl is not None:
            c = self._c(ctrl)
            dt = T(c.dt)
            if y0_alt is not None and c.accept:
                sel = 1
        else:
            dt = T(dt_host)
        y = _np(y0_alt if sel else y0).reshape(-1)
        kk = [_np(k0_alt if sel else ks[0]).reshape(-1)] + [_np(k).reshape(-1) for k in ks[1:]]
        o = _np(out).reshape(-1)
        with np.errstate(all="ignore"):
            if mode == _hip.COMBINE_RK:
                cs = [T(c_) * dt for c_ in coef]
                acc = kk[0] * cs[0]
                for j in range(1, len(kk)):
                    acc = acc + kk[j] * cs[j]
                if out2 is not None:
                    c2 = [dt * T(c_) for c_ in coef2]
                    e = kk[0] * c2[0]
                    for j in range(1, len(kk)):
                        e = e + kk[j] * c2[j]
                    _np(out2).reshape(-1)[...] = e
                o[...] = y + acc
            elif mode == _hip.COMBINE_FUSE:
                cs = [T(c_) for c_ in coef]
                acc = kk[0] * cs[0]
                for j in range(1, len(kk)):
                    acc = acc + kk[j] * cs[j]
                o[...] = self._fuse(acc, dt, y, T(damping))
                if out2 is not None:  # the leading terms of a later WFUSE launch: sum_j fuse(k_j) * w_j, left to right
                    c2 = [T(c_) for c_ in coef2]
                    e = self._fuse(kk[0], dt, y, T(damping)) * c2[0]
                    for j in range(1, len(kk)):
                        e = e + self._fuse(kk[j], dt, y, T(damping)) * c2[j]
                    _np(out2).reshape(-1)[...] = e
            else:
                cs = [T(c_) for c_ in coef]
                acc = self._fuse(kk[0], dt, y, T(damping)) * cs[0]
                for j in range(1, len(kk)):
                    acc = acc + self._fuse(kk[j], dt, y, T(damping)) * cs[j]
                o[...] = acc * T(scale)

    def stage_combine_pre(self, out, y0, pre, ks, coef, *, dt_host=0.0, ctrl=None, y0_alt=None, nt_mask=0):
        """Contract of xde_stage_combine_pre: out = y0 + ((pre + ks[0] c_0) + ks[1] c_1 ...), c_j = T(coef_j) * dt."""
        self.launches.append("combine")
        T = _NP[out.dtype]
        sel = 0
        if ctrl is not None:
            c = self._c(ctrl)
            dt = T(c.dt)
            if y0_alt is not None and c.accept:
                sel = 1
        else:
            dt = T(dt_host)
        y = _np(y0_alt if sel else y0).reshape(-1)
        with np.errstate(all="ignore"):
            acc = _np(pre).reshape(-1)
            for k, c_ in zip(ks, coef):
                acc = acc + _np(k).reshape(-1) * (T(c_) * dt)
            _np(out).reshape(-1)[...] = y + acc

    def stage_combine_pre_weighted(self, out, y0, pre, ks, coef, *, scale=1.0, dt_host=0.0, ctrl=None, damping=0.0):
        """Contract of xde_stage_combine_pre_weighted: out = ((pre + fuse(ks[0]) w_0) + ...) * scale."""
        self.launches.append("combine")
        T = _NP[out.dtype]
        dt = T(self._c(ctrl).dt) if ctrl is not None else T(dt_host)
        y = _np(y0).reshape(-1)
        with np.errstate(all="ignore"):
            acc = _np(pre).reshape(-1)
            for k, c_ in zip(ks, coef):
                acc = acc + self._fuse(_np(k).reshape(-1), dt, y, T(damping)) * T(c_)
            _np(out).reshape(-1)[...] = acc * T(scale)

    @staticmethod
    def _fuse(dy, dt, y0, lam):
        if lam == 0:
            return dy * dt + y0
        y = dy * dt + y0
        return (dy - lam * y) * dt + y0

    # -- K2 -------------------------------------------------------------------------------------
    @staticmethod
    def _seg_reduce(r, yv, segs, norm_kind):
        vals, nfs = [], []
        for i in range(segs.n_seg):
            s, l = segs.seg_start[i], segs.seg_len[i]
            rs = np.abs(r[s : s + l])
            if norm_kind == _hip.NORM_RMS:
                vals.append(float(np.sum((rs * rs).astype(np.float64))))
            else:
                vals.append(float(rs.max()) if l else 0.0)
            nfs.append(float(np.count_nonzero(~np.isfinite(yv[s : s + l]))))
        return vals, nfs

    def error_norm_partial(self, ks, c_err, y0, y1, rtol, atol, segs, norm_kind, ws, *, dt_host=0.0, ctrl=None,
                           y0_alt=None, k0_alt=None, e_pre=None):
        self.launches.append("errnorm")
        T = _NP[y0.dtype]
        sel = 0
        if ctrl is not None:
            c = self._c(ctrl)
            dt = T(c.dt)
            if y0_alt is not None and c.accept:
                sel = 1
        else:
            dt = T(dt_host)
        y0v = _np(y0_alt if sel else y0).reshape(-1)
        y1v = _np(y1).reshape(-1)
        k_first = ks[0] if (e_pre is not None or not sel) else k0_alt
        kk = [_np(k_first).reshape(-1)] + [_np(k).reshape(-1) for k in ks[1:]]
        with np.errstate(all="ignore"):
            cs = [dt * T(c_) for c_ in c_err]
            e = kk[0] * cs[0] if e_pre is None else _np(e_pre).reshape(-1) + kk[0] * cs[0]
            for j in range(1, len(kk)):
                e = e + kk[j] * cs[j]
            tol = T(atol) + T(rtol) * np.fmax(np.abs(y0v), np.abs(y1v))
            r = e / tol
        self._slots[0] = self._seg_reduce(r, y0v, segs, norm_kind) + (norm_kind, segs.n_seg)

    def error_norm_control(self, ks, c_err, y0, y1, segs, ws, ctrl, params, t_span_dev, step_t_dev, t_stage, *, y0_alt=None,
                           k0_alt=None, e_pre=None):
        self.error_norm_partial(ks, c_err, y0, y1, params.rtol, params.atol, segs, params.norm_kind, ws, ctrl=ctrl, y0_alt=y0_alt,
                                k0_alt=k0_alt, e_pre=e_pre)
        self.rk_control(ctrl, params, ws, None, t_span_dev, step_t_dev, t_stage)

    def error_ratio(self, out, ks, c_err, y0, y1, rtol, atol, *, dt_host=0.0, ctrl=None, y0_alt=None, k0_alt=None, nonfinite_out=None):
        self.launches.append("ratio")
        T = _NP[y0.dtype]
        sel = 0
        if ctrl is not None:
            c = self._c(ctrl)
            dt = T(c.dt)
            if y0_alt is not None and c.accept:
                sel = 1
        else:
            dt = T(dt_host)
        y0v, y1v = _np(y0_alt if sel else y0).reshape(-1), _np(y1).reshape(-1)
        kk = [_np(k0_alt if sel else ks[0]).reshape(-1)] + [_np(k).reshape(-1) for k in ks[1:]]
        with np.errstate(all="ignore"):
            cs = [dt * T(c_) for c_ in c_err]
            e = kk[0] * cs[0]
            for j in range(1, len(kk)):
                e = e + kk[j] * cs[j]
            _np(out).reshape(-1)[...] = e / (T(atol) + T(rtol) * np.fmax(np.abs(y0v), np.abs(y1v)))
        if nonfinite_out is not None:
            nonfinite_out += float(np.count_nonzero(~np.isfinite(y0v)))

    def scaled_norm_partial(self, a, b, y0, rtol, atol, segs, norm_kind, ws, slot):
        self.launches.append("scalednorm")
        T = _NP[y0.dtype]
        yv = _np(y0).reshape(-1)
        av = _np(a).reshape(-1)
        with np.errstate(all="ignore"):
            scale = T(atol) + np.abs(yv) * T(rtol)
            num = av - _np(b).reshape(-1) if b is not None else av
            r = num / scale
        self._slots[slot] = self._seg_reduce(r, yv, segs, norm_kind) + (norm_kind, segs.n_seg)

    def norm_finalize(self, ws, slot, sums):
        self.launches.append("finalize")
        vals, nfs, _, n_seg = self._slots[slot]
        s = sums.numpy()
        s[:] = 0.0
        s[:n_seg] = vals
        s[_hip.XDE_MAX_SEG : _hip.XDE_MAX_SEG + n_seg] = nfs

    @staticmethod
    def _norm_from_sums(vals, counts, norm_kind, state_dtype):
        rnd = (lambda x: float(np.float32(x))) if state_dtype == _hip.XDE_F32 else float
        ratio, per = 0.0, []
        with np.errstate(all="ignore"):
            for i, (v, n) in enumerate(zip(vals, counts)):
                if norm_kind == _hip.NORM_RMS:
                    r = rnd(np.sqrt(rnd(np.float64(v) / np.float64(n))))
                else:
                    r = rnd(v)
                r = abs(r)
                per.append(r)
                if i == 0:
                    ratio = r
                else:
                    ratio = ratio if ratio != ratio else (r if r != r else max(ratio, r))
        return ratio, per

    def norm_result(self, sums, seg_count, norm_kind, state_dtype, result):
        s = sums.numpy()
        ratio, _ = self._norm_from_sums(list(s[: len(seg_count)]), seg_count, norm_kind, state_dtype)
        result.numpy()[0] = ratio

    # -- K3 -------------------------------------------------------------------------------------
    @staticmethod
    def _plan_next(c, p, step_t, t_stage):
        TT = np.float32 if p.time_dtype == _hip.XDE_F32 else np.float64
        d = TT(p.direction)
        t0, dt = TT(c.t1), TT(c.dt)
        with np.errstate(all="ignore"):
            t1 = t0 + dt
            on = 0
            if p.n_step_t > 0 and step_t is not None:
                nxt = TT(step_t[c.next_step_index])
                if d * t0 < d * nxt < d * (t0 + dt):
                    on, t1 = 1, nxt
                    dt = t1 - t0
            c.on_step_t = on
            c.dt = float(dt)
            c.t_plan = float(t1)
            if not c.done:
                if not (d * (t0 + dt) > d * t0) and c.status == 0:
                    c.status = _hip.STATUS_DT_UNDERFLOW
                if c.steps_in_interval >= p.max_num_steps and c.status == 0:
                    c.status = _hip.STATUS_MAX_STEPS
            Y = np.float32 if p.state_dtype == _hip.XDE_F32 else np.float64
            ts = t_stage.numpy()
            for i in range(p.n_stage):
                a = Y(p.alpha[i])
                ts[i] = Y(t1) if a == 1.0 else Y(t0) + a * Y(dt)

    def initial_step(self, phase, res, hs, params, t_start, t_probe, ctrl):
        Y = np.float32 if params.state_dtype == _hip.XDE_F32 else np.float64
        r, h = res.numpy(), hs.numpy()
        if phase == 2:  # phase 0 with the start time in res[2]
            phase, t_start = 0, float(r[2])
        with np.errstate(all="ignore"):
            if phase == 0:
                d0, d1 = Y(abs(r[0])), Y(abs(r[1]))
                h0 = Y(1e-6) if (d0 < 1e-5 or d1 < 1e-5) else Y(0.01) * d0 / d1
                h0 = abs(h0)
                h[0], h[1], h[2] = d0, d1, h0
                hs0 = -h0 if params.direction < 0 else h0
                self._c(ctrl).dt = float(hs0)
                t0 = np.float32(t_start) if params.time_dtype == _hip.XDE_F32 else np.float64(t_start)
                t_probe.numpy()[...] = t0 + hs0
            else:
                h0, d1 = Y(h[2]), Y(h[1])
                d2 = abs(Y(r[0]) / h0)
                if d1 <= 1e-15 and d2 <= 1e-15:
                    h1 = max(Y(1e-6), h0 * Y(1e-3))
                else:
                    m = d2 if d2 > d1 else d1
                    h1 = Y((Y(0.01) / m) ** Y(1.0 / (params.order - 1.0 + 1.0)))
                h1 = abs(h1)
                first = np.fmin(Y(100.0) * h0, h1)
                h[3] = float(np.float32(first)) if params.time_dtype == _hip.XDE_F32 else float(first)

    def initial_step_fused(self, phase, a, b, y0, segs, hs, params, t_start, t_probe, ctrl, n_out=0, t_span_dev=None, step_t_dev=None,
                           t_stage=None, keep_seq=False):
        """Contract of xde_initial_step_fused: the separate calls, composed."""
        import torch

        if t_start != t_start:  # NaN: the start time is the first output time
            t_start = float(t_span_dev.numpy()[0])

        ws, sums = self.new_workspace(None), self.new_sums(None)
        counts = [params.seg_count[i] for i in range(params.n_seg)]
        res = torch.zeros(2, dtype=torch.float64)

        def norm(x, y, out):
            self.scaled_norm_partial(x, y, y0, params.rtol, params.atol, segs, params.norm_kind, ws, 0)
            self.norm_finalize(ws, 0, sums)
            self.norm_result(sums, counts, params.norm_kind, params.state_dtype, out)

        if phase == 0:
            norm(y0, None, res[0:1])
            norm(a, None, res[1:2])
            self.initial_step(0, res, hs, params, t_start, t_probe, ctrl)
        else:
            norm(a, b, res[0:1])
            self.initial_step(1, res, hs, params, t_start, None, ctrl)
            hs[4] = res[0]
            self.ctrl_init(ctrl, params, t_start, 0.0, n_out, t_span_dev, step_t_dev, t_stage, first_step_dev=hs[3:4])

    def scaled_norm2_partial(self, f0, y0, rtol, atol, segs, norm_kind, ws):
        """Contract of xde_scaled_norm2_partial: the two separate passes, slot 0 <- norm(y0/scale), slot 1 <- norm(f0/scale)."""
        self.scaled_norm_partial(y0, None, y0, rtol, atol, segs, norm_kind, ws, 0)
        self.scaled_norm_partial(f0, None, y0, rtol, atol, segs, norm_kind, ws, 1)
        self.launches.pop()  # (one launch)

    def initial_step_tail(self, phase, ws, hs, params, t_start, t_probe, ctrl, n_out=0, t_span_dev=None, step_t_dev=None, t_stage=None,
                          keep_seq=False):
        """Contract of xde_initial_step_tail: finalize + result (+ the scalar phase, + ctrl_init), composed."""
        import torch

        if t_start != t_start:  # NaN: the start time is the first output time
            t_start = float(t_span_dev.numpy()[0])
        sums = self.new_sums(None)
        counts = [params.seg_count[i] for i in range(params.n_seg)]
        res = torch.zeros(2, dtype=torch.float64)
        n0 = len(self.launches)
        for slot in ((0, 1) if phase == 0 else (0,)):
            self.norm_finalize(ws, slot, sums)
            self.norm_result(sums, counts, params.norm_kind, params.state_dtype, res[slot : slot + 1])
        del self.launches[n0:]
        self.launches.append("initial_step_tail")
        if phase == 0:
            self.initial_step(0, res, hs, params, t_start, t_probe, ctrl)
        else:
            self.initial_step(1, res, hs, params, t_start, None, ctrl)
            hs[4] = res[0]
            self.ctrl_init(ctrl, params, t_start, 0.0, n_out, t_span_dev, step_t_dev, t_stage, first_step_dev=hs[3:4])

    def ctrl_init(self, ctrl, params, t_start, first_step, n_out, t_span_dev, step_t_dev, t_stage, first_step_dev=None, keep_seq=False):
        c = self._c(ctrl)
        seq = c.seq
        if t_start != t_start:  # NaN: the start time is the first output time
            t_start = float(t_span_dev.numpy()[0])
        C.memset(C.addressof(c), 0, C.sizeof(c))
        if keep_seq:
            c.seq = seq
        c.t0 = c.t1 = float(t_start)
        if first_step_dev is not None:
            first_step = float(params.direction) * abs(float(first_step_dev.numpy()[0]))
        c.dt = float(first_step)
        if params.replay and params.n_replay > 0:
            c.dt = float((C.c_double * 2).from_address(params.replay)[0])
        c.n_out = n_out
        c.ratio_prev = 1e-4
        d = float(params.direction)
        ts = t_span_dev.numpy()
        e = 1
        while e < n_out and d * ts[e] <= d * t_start:
            e += 1
        c.next_out = c.out_begin = c.out_end = e
        c.done = 1 if e >= n_out else 0
        idx = 0
        if params.n_step_t > 0 and step_t_dev is not None:
            st = step_t_dev.numpy()
            while idx < params.n_step_t and d * st[idx] <= d * t_start:
                idx += 1
            idx = min(idx, params.n_step_t - 1)
        c.next_step_index = idx
        self._plan_next(c, params, None if step_t_dev is None else step_t_dev.numpy(), t_stage)

    def ctrl_retarget(self, ctrl, params, t_span_dev, n_out):
        c = self._c(ctrl)
        c.seq += 1
        d = float(params.direction)
        ts = t_span_dev.numpy()
        e = 0
        if c.accept and c.n_accept > 0:
            while e < n_out and d * ts[e] <= d * c.t1:
                e += 1
        c.n_out = n_out
        c.out_begin, c.out_end, c.next_out = 0, e, e
        c.done = 1 if e >= n_out else 0
        c.steps_in_interval = 0
        if c.status == _hip.STATUS_MAX_STEPS:
            c.status = _hip.STATUS_OK

    def rk_control(self, ctrl, params, ws, sums, t_span_dev, step_t_dev, t_stage):
        self._rk_control(ctrl, params, ws, sums, t_span_dev, step_t_dev, t_stage)
        if getattr(self, "_snaps", None) is not None:
            self._snaps.append(self.ctrl_read(ctrl))

    def _rk_control(self, ctrl, params, ws, sums, t_span_dev, step_t_dev, t_stage):
        self.launches.append("control")
        c, p = self._c(ctrl), params
        c.seq += 1
        if c.done:
            c.accept = 0
            c.out_begin = c.out_end = c.next_out
            return
        if sums is not None:
            s = sums.numpy()
            vals, nfs = list(s[: p.n_seg]), list(s[_hip.XDE_MAX_SEG : _hip.XDE_MAX_SEG + p.n_seg])
        else:
            vals, nfs, _, _ = self._slots[0]
        counts = [p.seg_count[i] for i in range(p.n_seg)]
        ratio, per = self._norm_from_sums(vals, counts, p.norm_kind, p.state_dtype)
        for i, r in enumerate(per):
            c.ratio_seg[i] = r
        nonfinite = float(sum(nfs))
        TT = np.float32 if p.time_dtype == _hip.XDE_F32 else np.float64
        d = TT(p.direction)
        t0, dt, t1 = TT(c.t1), TT(c.dt), TT(c.t_plan)
        min_step, max_step = TT(p.min_step), TT(p.max_step)
        if nonfinite > 0 and c.status == 0:
            c.status = _hip.STATUS_NONFINITE
        accept = 1 if ratio <= 1.0 else 0
        if d * dt > max_step:
            accept = 0
        if d * dt <= min_step:
            accept = 1
        with np.errstate(all="ignore"):
            if ratio == 0.0:
                dt_next = dt * TT(p.ifactor)
            else:
                dfactor = TT(1) if ratio < 1.0 else TT(p.dfactor)
                r = TT(ratio)
                if p.pi_controller:
                    beta = TT(p.pi_beta)
                    alpha = TT(1) / TT(p.order) - TT(0.75) * beta
                    prev = TT(max(c.ratio_prev, 1e-4))
                    factor = np.fmin(TT(p.ifactor), np.fmax(TT(p.safety) * prev**beta / r**alpha, dfactor))
                else:
                    exponent = TT(1) / TT(p.order)
                    factor = np.fmin(TT(p.ifactor), np.fmax(TT(p.safety) / r**exponent, dfactor))
                dt_next = dt * TT(factor)
            mag = d * dt_next
            if mag == mag:
                mag = min(max(mag, min_step), max_step)
                dt_next = d * mag
        if p.replay and c.n_steps < p.n_replay:  # prescribed step sequence (xde_ctrl_params_t.replay)
            tab = (C.c_double * (2 * p.n_replay)).from_address(p.replay)
            accept = 1 if tab[2 * c.n_steps + 1] != 0.0 else 0
            if c.n_steps + 1 < p.n_replay:
                dt_next = TT(tab[2 * (c.n_steps + 1)])
        c.n_steps += 1
        c.steps_in_interval += 1
        if accept:
            c.n_accept += 1
            if ratio == ratio:
                c.ratio_prev = ratio
        else:
            c.n_reject += 1
        c.sel_used = c.accept
        c.accept = accept
        c.ratio = ratio
        c.nonfinite = nonfinite
        c.t0 = float(t0)
        c.t1 = float(t1) if accept else float(t0)
        c.dt_last = float(dt)
        c.dt = float(dt_next)
        ts = t_span_dev.numpy()
        b = e = c.next_out
        if accept:
            while e < c.n_out and d * TT(ts[e]) <= d * t1:
                e += 1
        c.out_begin, c.out_end, c.next_out = b, e, e
        if e > b:
            c.steps_in_interval = 0
        c.done = 1 if e >= c.n_out else 0
        if accept and c.on_step_t and c.next_step_index != p.n_step_t - 1:
            c.next_step_index += 1
        self._plan_next(c, p, None if step_t_dev is None else step_t_dev.numpy(), t_stage)

    def ctrl_read(self, ctrl) -> XdeCtrl:
        return XdeCtrl.from_buffer_copy(ctrl.numpy().tobytes())

    def ctrl_read_async(self, ctrl):
        return self.ctrl_read(ctrl)

    def ctrl_peek_async(self, ctrl):
        return self.ctrl_read(ctrl)

    def ctrl_peek_result(self, handle):
        return handle

    def ctrl_wait(self, handle):
        return handle

    # -- K4 -------------------------------------------------------------------------------------
    def dense_eval(self, out_base, ks, mid, y0, y1, f1, ctrl, t_span_dev, time_dtype, *, y0_alt=None, k0_alt=None,
                   expect_step=-1):
        self.launches.append("dense")
        c = self._c(ctrl)
        if not c.accept or c.out_end <= c.out_begin:
            return
        if expect_step >= 0 and c.n_steps != expect_step:
            return
        T = _NP[y0.dtype]
        TT = np.float32 if time_dtype == _hip.XDE_F32 else np.float64
        sel = 1 if (y0_alt is not None and c.sel_used) else 0
        y0v = _np(y0_alt if sel else y0).reshape(-1)
        f0v = _np(k0_alt if sel else ks[0]).reshape(-1)
        y1v, f1v = _np(y1).reshape(-1), _np(f1).reshape(-1)
        kk = [f0v] + [_np(k).reshape(-1) for k in ks[1:]]
        dt = T(TT(c.dt_last))
        acc = kk[0] * (dt * T(mid[0]))
        for j in range(1, len(kk)):
            acc = acc + kk[j] * (dt * T(mid[j]))
        ymid = y0v + acc
        t0, t1 = TT(c.t0), TT(c.t1)
        ts = t_span_dev.numpy()
        out = _np(out_base).reshape(out_base.shape[0], -1)
        two, five, three, four = T(2), T(5), T(3), T(4)
        ca = two * dt * (f1v - f0v) - T(8) * (y1v + y0v) + T(16) * ymid
        cb = dt * (five * f0v - three * f1v) + T(18) * y0v + T(14) * y1v - T(32) * ymid
        cc = dt * (f1v - four * f0v) - T(11) * y0v - T(5) * y1v + T(16) * ymid
        cd = dt * f0v
        for r in range(c.out_begin, c.out_end):
            x = T((TT(ts[r]) - t0) / (t1 - t0))
            total = y0v + x * cd
            xp = x * x
            total = total + xp * cc
            xp = xp * x
            total = total + xp * cb
            xp = xp * x
            total = total + xp * ca
            out[r, :] = total

    def scale_fanout(self, outs, g, factors, dt_dev=None):
        self.launches.append("fanout")
        T = _NP[g.dtype]
        dt = T(dt_dev.item()) if dt_dev is not None else T(1)
        gv = _np(g).reshape(-1)
        for o, f in zip(outs, factors):
            _np(o).reshape(-1)[...] = gv * (T(f) * dt)

    def hermite_gather(self, val, der, his, his_t, lags):
        self.launches.append("hermite")
        T = _NP[his.dtype]
        h = _np(his)
        ts = _np(his_t)
        Tn, D = h.shape[-2], h.shape[-1]
        hh = h.reshape(-1, Tn, D)
        v, dv = _np(val).reshape(hh.shape[0], -1, D), _np(der).reshape(hh.shape[0], -1, D)

        def h_at(j):
            jj = j if j < Tn - 1 else Tn - 2
            return ts[jj + 1] - ts[jj]

        def ser(j):
            return hh[:, j if j < Tn else Tn - 1, :]

        def drv(j):
            jj = j if j < Tn - 1 else Tn - 2
            return (ser(jj + 1) - ser(jj)) / h_at(jj)

        for l, tau in enumerate(_np(lags).reshape(-1)):
            i = int(np.searchsorted(ts, tau, side="left")) - 1
            i = min(max(i, 0), Tn - 1)
            h1 = h_at(i)
            h2 = h_at(0) if i == 0 else h_at(i - 1)
            s = T((tau - ts[i]) / h1)
            p0, p1, d0, d1 = ser(i) / h1, ser(i + 1) / h2, drv(i), drv(i + 1)
            s2 = s * s
            s3 = s2 * s
            c0, c1, c2, c3 = T(2) * s3 - T(3) * s2 + T(1), T(-2) * s3 + T(3) * s2, s3 - T(2) * s2 + s, s3 - s2
            g0, g1 = T(6) * s2 - T(6) * s, T(-6) * s2 + T(6) * s
            g2, g3 = T(3) * s2 - T(4) * s + T(1), T(3) * s2 - T(2) * s
            v[:, l, :] = (((c0 * p0 + c1 * p1) + c2 * d0) + c3 * d1) * h1
            dv[:, l, :] = ((g0 * p0 + g1 * p1) + g2 * d0) + g3 * d1

    def history_gather(self, val, der, his, his_t, lags, method):
        """Contract of xde_history_gather: "cubic" = hermite_gather; "linear" / "bez" = weighted rows, the kernel's op order."""
        if method == "cubic":
            return self.hermite_gather(val, der, his, his_t, lags)
        self.launches.append("history_" + method)
        T = _NP[his.dtype]
        h, ts = _np(his), _np(his_t)
        Tn, D = h.shape[-2], h.shape[-1]
        hh = h.reshape(-1, Tn, D)
        v, dv = _np(val).reshape(hh.shape[0], -1, D), _np(der).reshape(hh.shape[0], -1, D)
        M, span = (2, 1) if method == "linear" else (4, 3)

        def scale1(j):
            jj = j if j < Tn - span else Tn - span - 1
            return ts[jj + span] - ts[jj]

        for l, tau in enumerate(_np(lags).reshape(-1)):
            i = min(max(int(np.searchsorted(ts, tau, side="left")) - 1, 0), Tn - 1)
            h1 = scale1(i)
            s = T((tau - ts[i]) / h1)
            if M == 2:
                w, g = [-s + T(1), s], [T(-1), T(1)]
            else:
                s2 = s * s
                s3 = s2 * s
                a, b = T(3) * s2, T(2) * s
                w = [((-s3 + T(3) * s2) - T(3) * s) + T(1), (T(3) * s3 - T(6) * s2) + T(3) * s, T(-3) * s3 + T(3) * s2, s3]
                g = [(-a + T(3) * b) - T(3), (T(3) * a - T(6) * b) + T(3), T(-3) * a + T(3) * b, a]
            av = ad = None
            for k in range(M):
                p = hh[:, min(i + k, Tn - 1), :] / scale1(max(i - k, 0))
                av = w[k] * p if av is None else av + w[k] * p
                ad = g[k] * p if ad is None else ad + g[k] * p
            v[:, l, :] = av * h1
            dv[:, l, :] = ad

    def lag_grad(self, grad_y, der):
        """Contract of xde_lag_grad: products in the state dtype, summed in fp64 over every axis but the lag axis."""
        import torch

        self.launches.append("lag_grad")
        g = (_np(grad_y) * _np(der)).astype(np.float64)
        axes = tuple(a for a in range(g.ndim) if a != g.ndim - 2)
        return torch.from_numpy(g.sum(axis=axes).astype(_NP[der.dtype]))

    def dense_commit(self, out_base, ks, mid, y0, y1, f1, ctrl, t_span_dev, time_dtype):
        self.dense_eval(out_base, ks, mid, y0, y1, f1, ctrl, t_span_dev, time_dtype)
        self.commit(ctrl, y0, y1, ks[0], f1)

    def commit(self, ctrl, y0_dst, y1_src, f0_dst, f1_src):
        self.launches.append("commit")
        if self._c(ctrl).accept:
            y0_dst.copy_(y1_src)
            f0_dst.copy_(f1_src)

    class _Replayable:
        def __init__(self, backend, body, ctrl, launches):
            self.backend, self.body, self.ctrl, self.launches = backend, body, ctrl, launches

        def replay(self):
            # one read handle (= a snapshot of the control block) per controller launch of the body, in order
            self.backend._snaps = snaps = []
            try:
                self.body()
            finally:
                self.backend._snaps = None
            assert len(snaps) == self.launches
            return snaps

    def capture(self, body, ctrl, launches=1):
        # the double executes ops immediately: "capture" records the callable, each replay runs it
        return NumpyDoubleBackend._Replayable(self, body, ctrl, launches)

    def prof_enable(self, on=True):
        pass

    def prof_collect(self):
        return {}
