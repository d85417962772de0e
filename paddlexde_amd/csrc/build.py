"""Build libxde_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m paddlexde_amd.csrc.build [--force]
    python -m paddlexde_amd.csrc.build --sanitize [pytest args ...]

One object per translation unit (compiled in parallel), then one link.  No relocatable device code is needed: device
functions shared between kernels live in the ``*_device.hpp`` / ``xde_reduce.hpp`` headers.

``--sanitize`` ("ASan host build"): the HOST half of the library — argument checks, segment maps, the event pool,
mailbox plumbing, hipGraph surgery — built with AddressSanitizer + UndefinedBehaviorSanitizer (``-fsanitize=address,undefined
-fno-gpu-sanitize``: device code is compiled as usual; GPU sanitizers are not available on this pool) into
``lib/libxde_hip_asan.so``, and ``tests/test_cabi.py`` (no GPU needed: every entry point's null / stale-mirror / bad-argument
paths) run against it with the ASan runtime preloaded.  A finding aborts the run; the log is kept under ``profiles/``.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = sorted(glob.glob(os.path.join(HERE, "xde_*.hip")))
HEADERS = sorted(glob.glob(os.path.join(HERE, "xde_*.hpp")))
OBJ_DIR = os.path.join(HERE, "build")
OUT_DIR = os.path.join(os.path.dirname(HERE), "lib")
OUT = os.path.join(OUT_DIR, "libxde_hip.so")
INCLUDE = os.path.join(ROOT, "include")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # element-wise results follow the reference's unfused op order
    "-fPIC",
]


def kernel_stamp(kernel="combine"):
    """12 hex digits identifying the SOURCES of one kernel family (bench.py stamps recorded counter passes with it: a recorded HBM-traffic
    figure is only reported for the kernel source it was measured on)."""
    import hashlib

    files = {"combine": ["xde_combine.hip", "xde_common.hpp"],
             "errnorm": ["xde_norm.hip", "xde_errnorm_device.hpp", "xde_reduce.hpp", "xde_common.hpp"]}[kernel]
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(HERE, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:12]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SOURCES + HEADERS + [os.path.join(INCLUDE, "xde_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


SANITIZE_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-fno-sanitize-recover=undefined"]
OUT_ASAN = os.path.join(OUT_DIR, "libxde_hip_asan.so")


def build(force=False, verbose=True, sanitize=False):
    out = OUT_ASAN if sanitize else OUT
    if not sanitize and not force and not needs_build():
        return OUT
    obj_dir = os.path.join(OBJ_DIR, "asan") if sanitize else OBJ_DIR
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = SANITIZE_FLAGS if sanitize else []

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + FLAGS + extra + ["-I", INCLUDE, "-I", HERE, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-fsanitize=address,undefined", "-shared-libsan"] if sanitize else []) + ["-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def asan_runtime():
    """clang's shared ASan runtime (preloaded into the un-instrumented python that dlopens the instrumented library)."""
    hipcc = os.environ.get("HIPCC", "hipcc")
    r = subprocess.run([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        if not hits:
            raise RuntimeError("libclang_rt.asan-x86_64.so not found")
        path = sorted(hits)[-1]
    return path


def sanitize_run(pytest_args=()):
    """Build the instrumented library and run the C-ABI suite against it in a child python (ASan runtime preloaded, the library path
    swapped in through the binding's own LIB_PATH).  Returns the child's exit code; its combined output goes to stdout."""
    out = build(force=True, sanitize=True)
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan_runtime() + ((":" + env["LD_PRELOAD"]) if env.get("LD_PRELOAD") else "")
    # (python itself leaks by design at exit; UBSan findings abort: -fno-sanitize-recover)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1:protect_shadow_gap=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    args = list(pytest_args) or [os.path.join(ROOT, "tests", "test_cabi.py"), "-q", "-p", "no:cacheprovider"]
    # (after the tests: the instrumented library, and the runtime that checks it, must be what the process really had mapped)
    code = ("import sys; sys.path.insert(0, {root!r}); import paddlexde_amd._hip as h; h.LIB_PATH = {lib!r}; import pytest; "
            "rc = int(pytest.main({args!r})); maps = open('/proc/self/maps').read(); "
            "ok = 'libxde_hip_asan.so' in maps and 'libclang_rt.asan' in maps; "
            "print('instrumented library and ASan runtime mapped:', ok, flush=True); sys.exit(rc if ok else (rc or 9))").format(
                root=ROOT, lib=out, args=args)
    print("sanitized library:", out, flush=True)
    print("LD_PRELOAD=" + env["LD_PRELOAD"], "ASAN_OPTIONS=" + env["ASAN_OPTIONS"], flush=True)
    return subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT).returncode


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        sys.exit(sanitize_run(sys.argv[sys.argv.index("--sanitize") + 1:]))
    build(force="--force" in sys.argv)
