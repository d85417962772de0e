# The GPU suite under non-default kernel policies (`gpurun -- "bash profiles/tools/policies.sh"`); results: profiles/r02_policies.txt
run() { echo "== $1"; env $1 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -6; }
run "XDE_CTRL_FLAGS=0 XDE_SINGLE_ELEMS=0"
run "XDE_FUSE_CONTROL=1"
run "XDE_GRID_BLOCKS=48"
run "XDE_GRID_BLOCKS=4096 XDE_NT=0"
run "XDE_HOST_FIRST_STEP=1 XDE_NO_POOL=1 XDE_NT=3 XDE_NT_BYTES=4096"
