# Round 4: the GPU suite under non-default kernel policies, this round's switches included, and a soak of the randomised sweeps
# (`gpurun -- "bash profiles/tools/policies_r04.sh"`); results: profiles/r04_policies.txt
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
# (a gpurun call is limited to ~19 minutes and one policy takes ~5: `policies_r04.sh A` runs the first two, `B` the others + the soak)
run() { echo "== $1"; env $1 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py --deselect tests/test_gpu_demo.py 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -8; }
PART=${1:-A}
{
if [ "$PART" = A ]; then
run "XDE_ERRNORM_PRE=0 XDE_FUSED_FIRST_STEP=0 XDE_CTRL_FLAGS=0"
run "XDE_SINGLE_ELEMS=0 XDE_FUSE_CONTROL=1 XDE_DENSE_GRID=48"
run "XDE_GRID_BLOCKS=48 XDE_STAGE_NT_MODE=all XDE_NORM_GRID=2048"
else
run "XDE_GRID_BLOCKS=4096 XDE_NT=0 XDE_NT_BYTES=4096 XDE_HOST_FIRST_STEP=1 XDE_NO_POOL=1"
echo "== XDE_SWEEP_SCALE=10 (the randomised sweeps, 10 x as many seeded blocks)"
XDE_SWEEP_SCALE=10 python3 -m pytest tests/test_gpu_odeint.py tests/test_gpu_kernels.py -m gpu -q -k "randomised or sweep" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -8
fi
} 2>&1 | tee gpurun_out/policies_r04_$PART.txt
