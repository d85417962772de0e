"""The C-ABI shared library loads on a box without a GPU and exports every symbol include/xde_hip.h declares;
the ctypes mirrors of the ABI structs have the header's layout.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from paddlexde_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xde_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xde_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_what_python_binds():
    assert _declared() == sorted(_hip.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    lib = _hip.load_library()
    for sym in _declared():
        assert hasattr(lib, sym), sym
    assert lib.xde_abi_version() == _hip.ABI_VERSION == 6


def test_struct_layouts_match():
    lib = _hip.load_library()
    assert lib.xde_sizeof_ctrl() == C.sizeof(_hip.XdeCtrl) == 288
    assert lib.xde_sizeof_ctrl_params() == C.sizeof(_hip.XdeCtrlParams)
    assert lib.xde_sizeof_segments() == C.sizeof(_hip.XdeSegments)
    assert _hip.XdeCtrlParams.struct_size.offset == 0 and _hip.XdeCtrlParams.abi_version.offset == 4 and _hip.XdeSegments.struct_size.offset == 0
    assert _hip.XdeCtrl.seq.offset % 8 == 0
    assert lib.xde_workspace_bytes() > 0
    # constants mirrored from the header
    src = open(HEADER).read()
    for name, val in [("XDE_MAX_K", _hip.XDE_MAX_K), ("XDE_MAX_SEG", _hip.XDE_MAX_SEG), ("XDE_MAX_STAGE", _hip.XDE_MAX_STAGE),
                      ("XDE_MIRROR_SLOTS", _hip.XDE_MIRROR_SLOTS)]:
        assert int(re.search(r"#define {}\s+(\d+)".format(name), src).group(1)) == val


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any HIP call (safe on a CPU-only box)."""
    lib = _hip.load_library()
    assert lib.xde_stage_combine(None, None, None, None, None, None, 1, 0, 1.0, 0.0, None, 8, 0, None, None, 0.0, 0, None) == _hip.XDE_EBADARG
    assert lib.xde_hermite_gather(None, None, None, None, None, 1, 4, 2, 3, 0, None) == _hip.XDE_EBADARG
    assert b"null pointer" in lib.xde_last_error()
    assert lib.xde_norm_finalize(None, 0, None, None) == _hip.XDE_EBADARG
    assert lib.xde_ctrl_wait(None, 0, 1.0, None) == _hip.XDE_EBADARG


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch

    from paddlexde_amd import Dopri5, RK4, odeint

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    y0 = torch.ones(4, 2)
    t = torch.tensor([0.0, 1.0])
    with pytest.raises(_hip.XdeError, match="no CPU path"):
        odeint(lambda t_, y: -y, y0, t, solver=Dopri5)
    with pytest.raises(_hip.XdeError, match="no CPU path"):
        odeint(lambda t_, y: -y, y0, t, solver=RK4)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libxde_hip.so")
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_backend", None)
    with pytest.raises(_hip.XdeError, match="is missing"):
        _hip.get_backend()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "paddlexde_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, flags=re.M), os.path.join(dirpath, f)
                assert "xde_oracle" not in src and "import_module(\"oracle" not in src, os.path.join(dirpath, f)


def test_every_entry_point_rejects_null_arguments():
    """Argument validation precedes any HIP call: with null pointers every compute entry point returns XDE_EBADARG and names
    itself in xde_last_error() — checked here without a GPU (nothing is launched)."""
    import ctypes as C

    from paddlexde_amd import _hip

    lib = _hip.load_library()
    P = _hip.XdeCtrlParams()
    S = _hip.XdeSegments()
    n = C.c_int(0)
    calls = {
        "xde_stage_combine": lambda: lib.xde_stage_combine(None, None, None, None, None, None, 1, 0, 1.0, 0.0, None, 8, 0, None, None, 0.0, 0, None),
        "xde_stage_combine_pre": lambda: lib.xde_stage_combine_pre(None, None, None, None, None, None, 1, 0.0, None, 8, 0, 0, None),
        "xde_stage_combine_pre_weighted": lambda: lib.xde_stage_combine_pre_weighted(None, None, None, None, None, 1, 0.125, 0.0, None, 8, 0, 0.0, None),
        "xde_error_norm_partial": lambda: lib.xde_error_norm_partial(None, None, None, 1, None, None, None, 1e-3, 1e-6, 0.0, None, C.byref(S), 0, 0,
                                                                    None, None, None),
        "xde_error_norm_control": lambda: lib.xde_error_norm_control(None, None, None, 1, None, None, None, C.byref(S), 0, None, None, None,
                                                                    C.byref(P), None, None, None, None, None),
        "xde_error_ratio": lambda: lib.xde_error_ratio(None, None, None, None, 1, None, None, None, 1e-3, 1e-6, 0.0, None, 8, 0, None, None),
        "xde_scaled_norm_partial": lambda: lib.xde_scaled_norm_partial(None, None, None, 1e-3, 1e-6, C.byref(S), 0, 0, None, 0, None),
        "xde_norm_finalize": lambda: lib.xde_norm_finalize(None, 0, None, None),
        "xde_norm_result": lambda: lib.xde_norm_result(None, None, 1, 0, 0, None, None),
        "xde_rk_control": lambda: lib.xde_rk_control(None, C.byref(P), None, None, None, None, None, None, None),
        "xde_ctrl_init": lambda: lib.xde_ctrl_init(None, C.byref(P), 0.0, 0.1, 2, None, None, None, 0, None, None),
        "xde_ctrl_retarget": lambda: lib.xde_ctrl_retarget(None, C.byref(P), None, 1, None, None),
        "xde_initial_step": lambda: lib.xde_initial_step(0, None, None, C.byref(P), 0.0, None, 0, None, None),
        "xde_initial_step_fused": lambda: lib.xde_initial_step_fused(0, None, None, None, C.byref(S), 0, None, C.byref(P), 0.0, None, 0, None, 2, None,
                                                                    None, None, 0, None),
        "xde_scaled_norm2_partial": lambda: lib.xde_scaled_norm2_partial(None, None, 1e-3, 1e-6, C.byref(S), 0, 0, None, None),
        "xde_initial_step_tail": lambda: lib.xde_initial_step_tail(0, None, None, C.byref(P), 0.0, None, 0, None, 2, None, None, None, 0, None, None),
        "xde_ctrl_read": lambda: lib.xde_ctrl_read(None, None, None),
        "xde_host_alloc": lambda: lib.xde_host_alloc(0, None),
        "xde_ctrl_wait": lambda: lib.xde_ctrl_wait(None, 0, 1.0, None),
        "xde_dense_eval": lambda: lib.xde_dense_eval(None, None, None, None, 1, None, None, None, None, None, None, 0, 8, 0, -1, None),
        "xde_dense_commit": lambda: lib.xde_dense_commit(None, None, None, 1, None, None, None, None, None, None, 0, 8, 0, None),
        "xde_commit": lambda: lib.xde_commit(None, None, None, None, None, 8, 0, None),
        "xde_pack_segments": lambda: lib.xde_pack_segments(None, None, None, None, None, 1, 8, 0, None),
        "xde_hermite_gather": lambda: lib.xde_hermite_gather(None, None, None, None, None, 1, 4, 2, 1, 0, None),
        "xde_history_gather": lambda: lib.xde_history_gather(None, None, None, None, None, 1, 4, 2, 1, 0, 1, None),
        "xde_lag_grad": lambda: lib.xde_lag_grad(None, None, None, 1, 4, 2, 0, None, None),
        "xde_scale_fanout": lambda: lib.xde_scale_fanout(None, None, None, 1, None, 8, 0, None),
        "xde_graph_replace_memsets": lambda: lib.xde_graph_replace_memsets(None, C.byref(n)),
        "xde_p2p_alloc": lambda: lib.xde_p2p_alloc(None),
        "xde_p2p_export": lambda: lib.xde_p2p_export(None, None),
        "xde_p2p_import": lambda: lib.xde_p2p_import(None, None),
        "xde_p2p_exchange": lambda: lib.xde_p2p_exchange(None, None, None, 2, 0, 0, 1000, None),
        "xde_p2p_error": lambda: lib.xde_p2p_error(None, None, None),
        "xde_p2p_error_info": lambda: lib.xde_p2p_error_info(None, None, 3, None),
        "xde_p2p_rk_control": lambda: lib.xde_p2p_rk_control(None, C.byref(P), None, None, None, 2, 0, 1000, None, None, None, None, None),
        "xde_prof_collect": lambda: lib.xde_prof_collect(None, None, None),
    }
    for name, call in calls.items():
        rc = call()
        msg = lib.xde_last_error().decode()
        assert rc == _hip.XDE_EBADARG, (name, rc, msg)
        assert msg and (name in msg or "segments" in msg), (name, msg)
    covered = set(calls) | {"xde_last_error", "xde_abi_version", "xde_sizeof_ctrl", "xde_sizeof_ctrl_params", "xde_sizeof_segments", "xde_workspace_bytes", "xde_host_free", "xde_prof_enable",
                                "xde_p2p_mailbox_bytes", "xde_p2p_free", "xde_p2p_close", "xde_lag_grad_workspace_bytes"}
    assert covered == set(_hip.SYMBOLS), set(_hip.SYMBOLS) ^ covered
    assert lib.xde_host_free(None) == _hip.XDE_OK  # freeing nothing is fine


def _old_params_mirror(short_by):
    """A binding's hand-written mirror of xde_ctrl_params_t from BEFORE a field was added: `short_by` bytes shorter than the
    library's struct (round 2's abort: a mirror without replay/n_replay, 16 bytes short, gpurun_out/r02_gpu1.log)."""
    real = C.sizeof(_hip.XdeCtrlParams)
    assert short_by % 8 == 0 and 0 < short_by < real

    class Old(C.Structure):
        _fields_ = [("words", C.c_double * ((real - short_by) // 8))]

    return Old


@pytest.mark.parametrize("stale", ["r02 layout (no size word, 16 bytes short)", "size word says 16 bytes short", "right size, old abi_version", "zeroed"])
def test_a_stale_params_mirror_gets_ebadarg_not_a_read_past_its_end(stale):
    """the library, not only the caller, guards the struct the round-2 abort came through.  Every entry point that
    takes xde_ctrl_params_t checks struct_size / abi_version — the first 8 bytes — before it reads anything else: a stale mirror
    yields XDE_EBADARG with a message that names the mismatch; nothing is launched (this runs on a box without a GPU)."""
    lib = _hip.load_library()
    real = C.sizeof(_hip.XdeCtrlParams)
    if stale.startswith("r02"):
        Old = _old_params_mirror(16 + 8)  # round 2's struct had neither the size word nor replay/n_replay
        P = Old()
        P.words[0], P.words[1] = 1e-5, 1e-7  # rtol, atol sat where struct_size/abi_version are now
    elif stale.startswith("size word"):
        P = _hip.XdeCtrlParams()
        P.struct_size = real - 16
    elif stale.startswith("right size"):
        P = _hip.XdeCtrlParams()
        P.abi_version = _hip.ABI_VERSION - 1
    else:
        P = _hip.XdeCtrlParams()
        C.memset(C.byref(P), 0, real)
    if isinstance(P, _hip.XdeCtrlParams):  # otherwise plausible, so that only the layout words can be what is refused
        P.n_stage, P.n_seg, P.direction, P.order = 6, 1, 1, 5.0
    dummy = C.c_void_p(0x1000)  # non-null "device pointers": validation must stop before anything dereferences or launches
    S = _hip.XdeSegments()
    S.n_seg = 1
    ref = C.cast(C.byref(P), C.POINTER(_hip.XdeCtrlParams))
    kk = (C.c_void_p * 1)(0x1000)
    ce = (C.c_double * 1)(1.0)
    calls = {
        "xde_ctrl_init": lambda: lib.xde_ctrl_init(dummy, ref, 0.0, 0.1, 2, dummy, None, dummy, 0, None, None),
        "xde_rk_control": lambda: lib.xde_rk_control(dummy, ref, dummy, None, dummy, None, dummy, None, None),
        "xde_ctrl_retarget": lambda: lib.xde_ctrl_retarget(dummy, ref, dummy, 1, None, None),
        "xde_initial_step": lambda: lib.xde_initial_step(1, dummy, dummy, ref, 0.0, None, 0, dummy, None),
        "xde_initial_step_tail": lambda: lib.xde_initial_step_tail(1, dummy, dummy, ref, 0.0, None, 0, dummy, 2, dummy, None, dummy, 0, None, None),
        "xde_error_norm_control": lambda: lib.xde_error_norm_control(kk, None, ce, 1, dummy, None, dummy, C.byref(S), 0, dummy, None, dummy,
                                                                    ref, dummy, None, dummy, None, None),
    }
    for name, call in calls.items():
        assert call() == _hip.XDE_EBADARG, name
        msg = lib.xde_last_error().decode()
        assert name in msg and "layout mismatch" in msg and "sizeof={}".format(real) in msg, msg


def test_a_stale_segments_mirror_gets_ebadarg():
    lib = _hip.load_library()

    class OldSegments(C.Structure):  # round 2's xde_segments_t: n_seg first, no size word
        _fields_ = [("n_seg", C.c_int32), ("seg_start", C.c_int64 * _hip.XDE_MAX_SEG), ("seg_len", C.c_int64 * _hip.XDE_MAX_SEG)]

    S = OldSegments()
    S.n_seg = 1
    S.seg_len[0] = 1024
    dummy = C.c_void_p(0x1000)
    kk = (C.c_void_p * 1)(0x1000)
    ce = (C.c_double * 1)(1.0)
    sref = C.cast(C.byref(S), C.POINTER(_hip.XdeSegments))
    assert lib.xde_scaled_norm_partial(dummy, None, dummy, 1e-3, 1e-6, sref, 0, 0, dummy, 0, None) == _hip.XDE_EBADARG
    assert "xde_segments_t layout mismatch" in lib.xde_last_error().decode()
    assert lib.xde_error_norm_partial(kk, None, ce, 1, dummy, None, dummy, 1e-3, 1e-6, 0.1, None, sref, 0, 0, dummy, None, None) == _hip.XDE_EBADARG
    assert "xde_segments_t layout mismatch" in lib.xde_last_error().decode()


def test_header_is_plain_c_and_the_cpp_example_builds(tmp_path):
    """include/xde_hip.h is the boundary a foreign host binds: it must be valid C (and C++), and examples/cabi_dopri5.cpp — a whole
    solve from plain C++/HIP — must build against it and the in-tree library (hipcc cross-compiles without a GPU; it RUNS in the GPU
    suite)."""
    import shutil
    import subprocess

    for cc, lang, std in (("gcc", "c", "c99"), ("g++", "c++", "c++11")):
        r = subprocess.run([cc, "-fsyntax-only", "-x", lang, "-std=" + std, "-Wall", "-Wextra", "-Werror", HEADER], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir = os.path.join(ROOT, "paddlexde_amd", "lib")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "cabi_dopri5.cpp"), "-L", lib_dir, "-lxde_hip", "-o", str(tmp_path / "cabi_dopri5")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_host_halves_run_up_to_the_launch_without_a_gpu():
    """SURVEY section 5 ("ASan host build") / VERDICT r04 (missing 4): everything an entry point does on the HOST before its launch —
    argument blocks, alignment and vector-path decisions, segment maps with their block shares, grid sizes, cache-policy selection,
    profiling scopes — executed with well-formed arguments on a box WITHOUT a GPU: the launch itself then fails (no device) and the call
    returns XDE_EHIP with the HIP error's text, never touching the (fake, never dereferenced on the host) device addresses.  Under
    `python -m paddlexde_amd.csrc.build --sanitize` these paths run instrumented (profiles/r05_sanitize_cabi.txt)."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the launches would run on the fake addresses")
    lib = _hip.load_library()
    vp = C.c_void_p
    dev = lambda i: 0x7F0000000000 + 0x100000 * i  # noqa: E731  16-byte aligned "device" addresses
    P = _hip.XdeCtrlParams()
    P.rtol, P.atol, P.safety, P.ifactor, P.dfactor, P.order = 1e-5, 1e-7, 0.9, 10.0, 0.2, 5.0
    P.max_step, P.max_num_steps, P.direction, P.n_stage, P.n_seg = float("inf"), 100, 1, 6, 3
    segs = _hip.make_segments([(0, 4), (4, 1000), (1004, 36)])
    for i, (_, n_) in enumerate([(0, 4), (4, 1000), (1004, 36)]):
        P.seg_count[i] = float(n_)
    n = 1040
    ks = (vp * 7)(*[dev(10 + j) for j in range(7)])
    coef = (C.c_double * 7)(*[0.1 * (j + 1) for j in range(7)])
    outs = (vp * 3)(dev(30), dev(31), dev(32))
    lens = (C.c_int64 * 3)(4, 1000, 36)
    starts = (C.c_int64 * 3)(0, 4, 1004)
    calls = {
        "stage_combine vec": lambda: lib.xde_stage_combine(dev(1), dev(2), dev(3), ks, dev(4), coef, 5, 0, 1.0, 0.0, dev(5), n, 0, dev(6), coef, 0.0, 0x1F, None),
        "stage_combine scalar path": lambda: lib.xde_stage_combine(dev(1) + 4, dev(2), None, ks, None, coef, 7, 2, 0.125, 0.01, None, n + 3, 1, None, None, 0.001, 0, None),
        "stage_combine 14 operands": lambda: lib.xde_stage_combine(dev(1), dev(2), None, (vp * 14)(*[dev(40 + j) for j in range(14)]), None,
                                                                  (C.c_double * 14)(*([0.5] * 14)), 14, 0, 1.0, 0.1, None, n, 0, None, None, 0.0, 0, None),
        "stage_combine_pre": lambda: lib.xde_stage_combine_pre(dev(1), dev(2), dev(3), dev(7), ks, coef, 1, 0.0, dev(5), n, 0, 1, None),
        "stage_combine FUSE + second output": lambda: lib.xde_stage_combine(dev(1), dev(2), None, ks, None, coef, 3, 1, 1.0, 0.01, None, n, 0, dev(6), coef, 0.001, 0, None),
        "stage_combine_pre_weighted": lambda: lib.xde_stage_combine_pre_weighted(dev(1), dev(2), dev(7), ks, coef, 1, 0.125, 0.01, None, n, 1, 0.001, None),
        "error_norm_partial 3 segments": lambda: lib.xde_error_norm_partial(ks, None, coef, 6, dev(2), None, dev(8), 1e-5, 1e-7, 0.01, None, C.byref(segs), 0, 0,
                                                                           dev(9), None, None),
        "error_norm_partial fsal": lambda: lib.xde_error_norm_partial(ks, None, coef, 1, dev(2), dev(3), dev(8), 1e-5, 1e-7, 0.0, dev(5), C.byref(segs), 1, 1,
                                                                     dev(9), dev(6), None),
        "error_norm_control": lambda: lib.xde_error_norm_control(ks, None, coef, 1, dev(2), None, dev(8), C.byref(segs), 0, dev(9), dev(6), dev(5), C.byref(P),
                                                                dev(20), None, dev(21), None, None),
        "scaled_norm_partial": lambda: lib.xde_scaled_norm_partial(dev(1), dev(2), dev(3), 1e-5, 1e-7, C.byref(segs), 0, 0, dev(9), 1, None),
        "scaled_norm2_partial": lambda: lib.xde_scaled_norm2_partial(dev(1), dev(2), 1e-5, 1e-7, C.byref(segs), 0, 0, dev(9), None),
        "initial_step_tail": lambda: lib.xde_initial_step_tail(1, dev(9), dev(23), C.byref(P), 0.0, None, 0, dev(5), 2, dev(20), None, dev(21), 0, None, None),
        "norm_finalize": lambda: lib.xde_norm_finalize(dev(9), 0, dev(22), None),
        "rk_control": lambda: lib.xde_rk_control(dev(5), C.byref(P), dev(9), None, dev(20), None, dev(21), None, None),
        "ctrl_init": lambda: lib.xde_ctrl_init(dev(5), C.byref(P), 0.0, 0.1, 2, dev(20), None, dev(21), 0, None, None),
        "initial_step_fused": lambda: lib.xde_initial_step_fused(0, dev(1), None, dev(2), C.byref(segs), 0, dev(23), C.byref(P), 0.0, dev(24), 0, dev(5), 2, dev(20),
                                                                None, dev(21), 0, None),
        "dense_eval": lambda: lib.xde_dense_eval(dev(25), ks, None, coef, 6, dev(2), None, dev(8), dev(16), dev(5), dev(20), 0, n, 0, -1, None),
        "pack_segments": lambda: lib.xde_pack_segments(dev(1), outs, starts, lens, None, 3, n, 0, None),
        "history_gather bez": lambda: lib.xde_history_gather(dev(1), dev(2), dev(3), dev(4), dev(6), 100, 24, 64, 12, 0, 2, None),
        "lag_grad plane": lambda: lib.xde_lag_grad(dev(1), dev(2), dev(3), 9824, 64, 12, 0, dev(9), None),
        "lag_grad long rows": lambda: lib.xde_lag_grad(dev(1), dev(2), dev(3), 9, 4100, 3, 1, dev(9), None),
        "lag_grad unaligned": lambda: lib.xde_lag_grad(dev(1), dev(2) + 8, dev(3), 50, 700, 3, 0, dev(9), None),
        "scale_fanout": lambda: lib.xde_scale_fanout(outs, dev(1), coef, 3, None, n, 1, None),
        "commit": lambda: lib.xde_commit(dev(5), dev(1), dev(2), dev(3), dev(4), n, 0, None),
    }
    for name, call in calls.items():
        rc = call()
        msg = lib.xde_last_error().decode()
        assert rc == _hip.XDE_EHIP, (name, rc, msg)  # the arguments were fine; only the device is missing
        assert msg, name
    # with kernel timing on, the profiling scope's event bookkeeping runs too (events cannot be created without a device: refused, not crashed)
    assert lib.xde_prof_enable(1) in (_hip.XDE_OK, _hip.XDE_EHIP)
    lib.xde_stage_combine(dev(1), dev(2), None, ks, None, coef, 2, 0, 1.0, 0.1, None, n, 0, None, None, 0.0, 0, None)
    assert lib.xde_prof_enable(0) in (_hip.XDE_OK, _hip.XDE_EHIP)


def _ctrl_checksum(block):
    """Python statement of the control block's checksum (csrc/xde_control_device.hpp: ctrl_chk_term / ctrl_mix64): the sum over every
    8-byte word but `chk` of splitmix64's finaliser of (word + golden * (index + 1)), mod 2^64."""
    M = (1 << 64) - 1

    def mix(x):
        x ^= x >> 30
        x = (x * 0xBF58476D1CE4E5B9) & M
        x ^= x >> 27
        x = (x * 0x94D049BB133111EB) & M
        x ^= x >> 31
        return x

    words = (C.c_uint64 * (C.sizeof(_hip.XdeCtrl) // 8)).from_buffer_copy(bytes(block))
    chk_word = _hip.XdeCtrl.chk.offset // 8
    return sum(mix((w + 0x9E3779B97F4A7C15 * (i + 1)) & M) for i, w in enumerate(words) if i != chk_word) & M


def test_host_mirror_accepts_only_whole_blocks(monkeypatch):
    """Round 5: the controller publishes its block to the pinned host mirror with ONE unordered store instruction; `xde_ctrl_wait`
    accepts a copy of the slot only if its sequence number is the expected one AND its checksum holds.  Checked here on a mirror ring
    in ordinary host memory (no GPU): a whole block is returned; a slot in which one payload word is still the previous occupant's (a
    copy taken while the words were landing) is NOT accepted — the wait runs into its timeout instead of handing back a torn block;
    a slot already taken by a later launch is reported as such."""
    if "XDE_CTRL_FLAGS" in os.environ and not int(os.environ["XDE_CTRL_FLAGS"]) & 8:
        pytest.skip("the seqlock protocol is selected")
    lib = _hip.load_library()
    ring = (_hip.XdeCtrl * _hip.XDE_MIRROR_SLOTS)()
    seq = 21
    new, old = _hip.XdeCtrl(), _hip.XdeCtrl()
    for blk, s in ((old, seq - _hip.XDE_MIRROR_SLOTS), (new, seq)):
        blk.t0, blk.t1, blk.dt, blk.ratio = 0.1 * s, 0.1 * s + 0.05, 0.05 + 1e-3 * s, 0.5
        blk.n_steps, blk.n_accept, blk.accept, blk.seq = s, s - 2, 1, s
        blk.chk = _ctrl_checksum(blk)
    out = _hip.XdeCtrl()
    slot = seq % _hip.XDE_MIRROR_SLOTS
    ring[slot] = new
    assert lib.xde_ctrl_wait(ring, seq, 50.0, C.byref(out)) == _hip.XDE_OK
    assert (out.seq, out.dt, out.n_steps, out.chk) == (seq, new.dt, seq, new.chk)
    # torn: every word of the new block has landed except `dt`, which is still the block's that held this slot 16 launches ago
    torn = _hip.XdeCtrl.from_buffer_copy(bytes(new))
    torn.dt = old.dt
    ring[slot] = torn
    assert lib.xde_ctrl_wait(ring, seq, 30.0, C.byref(out)) == _hip.XDE_ETIMEOUT
    # (a rejected candidate never reaches the caller's buffer — it still holds the block of the last successful wait)
    assert bytes(out) == bytes(new)
    # ... and so is a slot whose sequence number has landed while the rest is still the old block
    torn = _hip.XdeCtrl.from_buffer_copy(bytes(old))
    torn.seq = seq
    ring[slot] = torn
    assert lib.xde_ctrl_wait(ring, seq, 30.0, C.byref(out)) == _hip.XDE_ETIMEOUT
    # the slot belongs to a later launch already: the host lagged a whole ring behind
    later = _hip.XdeCtrl.from_buffer_copy(bytes(new))
    later.seq = seq + _hip.XDE_MIRROR_SLOTS
    later.chk = _ctrl_checksum(later)
    ring[slot] = later
    assert lib.xde_ctrl_wait(ring, seq, 30.0, C.byref(out)) == _hip.XDE_EBADARG and b"overwritten" in lib.xde_last_error()
    assert bytes(out) == bytes(new)
