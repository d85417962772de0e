# round 5, call j: final GPU suite on the committed tree + a 6x soak of the randomised sweeps (new combine kernels, RK4 pre-summed final sum,
# lag_grad mappings) + the suite's kernel-level tests under two non-default policies (seqlock mirror publish; tiny grids)
mkdir -p gpurun_out/r05j
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05j/suite.log 2>&1; echo rc=$? >> gpurun_out/r05j/suite.log; tail -3 gpurun_out/r05j/suite.log
XDE_SWEEP_SCALE=6 timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "randomised" > gpurun_out/r05j/soak.log 2>&1; echo rc=$? >> gpurun_out/r05j/soak.log; tail -3 gpurun_out/r05j/soak.log
XDE_CTRL_FLAGS=0 XDE_GRID_BLOCKS=48 XDE_NORM_GRID=48 XDE_LAG_GRID=7 timeout -k 10 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not bench and not sharded_gloo and not demo" > gpurun_out/r05j/policy_a.log 2>&1; echo rc=$? >> gpurun_out/r05j/policy_a.log; tail -3 gpurun_out/r05j/policy_a.log
XDE_CTRL_FLAGS=9 XDE_GRID_BLOCKS=4096 XDE_NT=0 XDE_SINGLE_ELEMS=0 XDE_LAG_GRID=2048 timeout -k 10 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not bench and not sharded_gloo and not demo" > gpurun_out/r05j/policy_b.log 2>&1; echo rc=$? >> gpurun_out/r05j/policy_b.log; tail -3 gpurun_out/r05j/policy_b.log
