"""Build libxde_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m paddlexde_amd.csrc.build
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "xde_hip.hip")
OUT_DIR = os.path.join(os.path.dirname(HERE), "lib")
OUT = os.path.join(OUT_DIR, "libxde_hip.so")
INCLUDE = os.path.join(ROOT, "include")

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # element-wise results follow the reference's unfused op order
    "-fPIC",
    "-shared",
]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SRC, os.path.join(INCLUDE, "xde_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + ["-I", INCLUDE, "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
