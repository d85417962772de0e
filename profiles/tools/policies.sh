# The GPU suite under non-default kernel policies (`gpurun -- "bash profiles/tools/policies.sh"`); results: profiles/rNN_policies.txt
mkdir -p gpurun_out
run() { echo "== $1"; env $1 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py --deselect tests/test_gpu_demo.py 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -6; }
{
run "XDE_CTRL_FLAGS=0 XDE_SINGLE_ELEMS=0"
run "XDE_FUSE_CONTROL=1 XDE_DENSE_GRID=48"
run "XDE_GRID_BLOCKS=48 XDE_STAGE_NT_MODE=all"
run "XDE_GRID_BLOCKS=4096 XDE_NT=0 XDE_DENSE_GRID=2048"
run "XDE_HOST_FIRST_STEP=1 XDE_NO_POOL=1 XDE_NT=3 XDE_NT_BYTES=4096 XDE_STAGE_NT_MODE=old"
} 2>&1 | tee gpurun_out/policies.txt
