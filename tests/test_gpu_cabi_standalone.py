"""The C ABI on its own: a complete adaptive Dopri5 integration driven with raw ctypes calls into libxde_hip.so
(device pointers + sizes + stream, as a foreign host framework — e.g. PaddlePaddle, see INTEGRATION.md — would bind
it), without the paddlexde_amd Python layer.  Only the structs' layouts are taken from include/xde_hip.h (mirrored
here by hand on purpose).  Result must equal paddlexde_amd.odeint bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_K, MAX_SEG, MAX_STAGE = 14, 16, 13


class Ctrl(C.Structure):  # xde_ctrl_t
    _fields_ = [(n, C.c_double) for n in ("t0", "t1", "dt", "dt_last", "t_plan", "ratio_prev", "ratio")] + [
        ("ratio_seg", C.c_double * MAX_SEG), ("nonfinite", C.c_double),
        ("n_steps", C.c_int64), ("n_accept", C.c_int64), ("n_reject", C.c_int64), ("steps_in_interval", C.c_int64)] + [
        (n, C.c_int32) for n in ("accept", "sel_used", "status", "out_begin", "out_end", "next_out", "n_out", "done",
                                 "next_step_index", "on_step_t")] + [("seq", C.c_int64), ("reserved", C.c_int32 * 4)]


class Params(C.Structure):  # xde_ctrl_params_t
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32)] + [
        (n, C.c_double) for n in ("rtol", "atol", "min_step", "max_step", "safety", "ifactor", "dfactor", "order")] + [
        ("max_num_steps", C.c_int64)] + [(n, C.c_int32) for n in ("time_dtype", "state_dtype", "direction", "norm_kind", "n_stage",
                                                                 "n_seg", "n_step_t", "pi_controller")] + [
        ("pi_beta", C.c_double), ("alpha", C.c_double * MAX_STAGE), ("seg_count", C.c_double * MAX_SEG),
        ("replay", C.c_void_p), ("n_replay", C.c_int64)]


class Segs(C.Structure):  # xde_segments_t
    _fields_ = [("struct_size", C.c_uint32), ("n_seg", C.c_int32), ("seg_start", C.c_int64 * MAX_SEG), ("seg_len", C.c_int64 * MAX_SEG)]


def test_dopri5_through_raw_c_abi():
    lib = C.CDLL(os.path.join(ROOT, "paddlexde_amd", "lib", "libxde_hip.so"))
    vp, dbl, i64, i32 = C.c_void_p, C.c_double, C.c_int64, C.c_int
    dp, vpp = C.POINTER(C.c_double), C.POINTER(C.c_void_p)
    lib.xde_last_error.restype = C.c_char_p
    lib.xde_sizeof_ctrl.restype = i64
    lib.xde_workspace_bytes.restype = i64
    lib.xde_stage_combine.argtypes = [vp, vp, vp, vpp, vp, dp, i32, i32, dbl, dbl, vp, i64, i32, vp, dp, dbl, C.c_uint32, vp]
    lib.xde_error_norm_partial.argtypes = [vpp, vp, dp, i32, vp, vp, vp, dbl, dbl, dbl, vp, C.POINTER(Segs), i32, i32, vp, vp, vp]
    lib.xde_rk_control.argtypes = [vp, C.POINTER(Params), vp, vp, vp, vp, vp, vp, vp]
    lib.xde_ctrl_init.argtypes = [vp, C.POINTER(Params), dbl, dbl, C.c_int32, vp, vp, vp, i64, vp, vp]
    lib.xde_ctrl_read.argtypes = [vp, C.POINTER(Ctrl), vp]
    lib.xde_dense_eval.argtypes = [vp, vpp, vp, dp, i32, vp, vp, vp, vp, vp, vp, i32, i64, i32, i64, vp]
    lib.xde_sizeof_ctrl_params.restype = i64
    assert lib.xde_abi_version() == 6
    assert lib.xde_sizeof_ctrl() == C.sizeof(Ctrl)
    assert lib.xde_sizeof_ctrl_params() == C.sizeof(Params)  # a hand-written mirror must be checked before it is passed

    def ok(rc):
        assert rc == 0, lib.xde_last_error().decode()

    dev = torch.device("cuda:0")
    B, D = 512, 64
    g = torch.Generator().manual_seed(1)
    U = 0.1 * torch.randn(D, D, generator=g)
    A = (U - U.T).to(dev)
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0)).to(dev)
    func = lambda t, y: y @ A.T  # noqa: E731  (the host framework's call)
    t_span = [0.0, 0.4, 1.0]
    n = y0.numel()
    stream = torch.cuda.current_stream().cuda_stream

    # Dormand-Prince tableau (reference: paddlexde/solver/adaptive_solver/dopri5.py:5-55)
    alpha = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
    beta = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
            [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656], [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
    c_err = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
             11 / 84 - 649 / 6300, -1.0 / 60.0]
    mid = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]

    p = Params()
    # the binding states the layout it was written against; a stale mirror gets XDE_EBADARG from every entry point (first, prove it)
    p.struct_size, p.abi_version = C.sizeof(Params) - 16, 6
    assert lib.xde_ctrl_init(0x1000, C.byref(p), 0.0, 0.01, 3, 0x1000, None, 0x1000, 0, None, stream) == 1
    assert "layout mismatch" in lib.xde_last_error().decode()
    p.struct_size, p.abi_version = C.sizeof(Params), 6
    p.rtol, p.atol, p.min_step, p.max_step = float(np.float32(1e-5)), float(np.float32(1e-7)), 0.0, float("inf")
    p.safety, p.ifactor, p.dfactor, p.order = float(np.float32(0.9)), 10.0, float(np.float32(0.2)), 5.0
    p.max_num_steps = 2**31 - 1
    p.time_dtype = p.state_dtype = 0
    p.direction, p.norm_kind, p.n_stage, p.n_seg = 1, 0, 6, 1
    for i, a in enumerate(alpha):
        p.alpha[i] = a
    p.seg_count[0] = float(n)
    segs = Segs()
    segs.struct_size = C.sizeof(Segs)
    segs.n_seg, segs.seg_start[0], segs.seg_len[0] = 1, 0, n

    ctrl = torch.zeros(C.sizeof(Ctrl), dtype=torch.uint8, device=dev)
    ws = torch.zeros(lib.xde_workspace_bytes(), dtype=torch.uint8, device=dev)
    t_dev = torch.tensor(t_span, dtype=torch.float64, device=dev)
    t_stage = torch.zeros(MAX_STAGE, dtype=torch.float32, device=dev)
    sol = torch.empty(len(t_span), B, D, device=dev)
    sol[0] = y0
    first_step = 0.01
    ok(lib.xde_ctrl_init(ctrl.data_ptr(), C.byref(p), 0.0, first_step, len(t_span), t_dev.data_ptr(), None, t_stage.data_ptr(), 0, None, stream))

    def ptrs(ts):
        return (C.c_void_p * len(ts))(*[x.data_ptr() for x in ts])

    def dbls(xs):
        return (C.c_double * len(xs))(*xs)

    y, f = y0, func(None, y0)
    host = Ctrl()
    for _ in range(1000):
        ks = [f]
        y1 = None
        for i in range(6):
            idx = [0] + [j for j in range(1, i + 1) if beta[i][j] != 0]
            out = torch.empty_like(y)
            ok(lib.xde_stage_combine(out.data_ptr(), y.data_ptr(), None, ptrs([ks[j] for j in idx]), None, dbls([beta[i][j] for j in idx]),
                                     len(idx), 0, 1.0, 0.0, ctrl.data_ptr(), n, 0, None, None, 0.0, 0, stream))
            ks.append(func(t_stage[i], out).contiguous())
            y1 = out
        eidx = [0, 2, 3, 4, 5, 6]
        ok(lib.xde_error_norm_partial(ptrs([ks[j] for j in eidx]), None, dbls([c_err[j] for j in eidx]), len(eidx), y.data_ptr(), None,
                                      y1.data_ptr(), p.rtol, p.atol, 0.0, ctrl.data_ptr(), C.byref(segs), 0, 0, ws.data_ptr(), None, stream))
        ok(lib.xde_rk_control(ctrl.data_ptr(), C.byref(p), ws.data_ptr(), None, t_dev.data_ptr(), None, t_stage.data_ptr(), None, stream))
        ok(lib.xde_ctrl_read(ctrl.data_ptr(), C.byref(host), stream))
        assert host.status == 0
        if host.accept:
            if host.out_end > host.out_begin:
                midx = [0, 2, 3, 4, 5, 6]
                ok(lib.xde_dense_eval(sol.data_ptr(), ptrs([ks[j] for j in midx]), None, dbls([mid[j] for j in midx]), len(midx), y.data_ptr(),
                                      None, y1.data_ptr(), ks[-1].data_ptr(), ctrl.data_ptr(), t_dev.data_ptr(), 0, n, 0, -1, stream))
            y, f = y1, ks[-1]
        if host.done:
            break
    assert host.done and host.n_accept > 3

    from paddlexde_amd import Dopri5, odeint
    from paddlexde_amd.utils import _rms_norm

    ref = odeint(func, y0, torch.tensor(t_span), solver=Dopri5, rtol=1e-5, atol=1e-7, options={"norm": _rms_norm, "first_step": first_step})
    assert torch.equal(sol, ref)


def test_cpp_host_program_through_the_c_abi(tmp_path):
    """examples/cabi_dopri5.cpp: a complete adaptive Dopri5 solve driven from plain C++/HIP through libxde_hip.so — hipMalloc'd buffers,
    a HIP kernel as the user's func, no Python and no torch in the process.  Built here with hipcc against include/xde_hip.h and run;
    it checks its result against the closed form and exits 0."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "cabi_dopri5")
    lib_dir = os.path.join(ROOT, "paddlexde_amd", "lib")
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "cabi_dopri5.cpp"), "-L", lib_dir, "-lxde_hip", "-Wl,-rpath," + lib_dir, "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr[-2000:])
    assert "accepted" in run.stdout and "worst |error|" in run.stdout, run.stdout
