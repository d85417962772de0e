# round 5, call k: where a step's time is — per-kernel durations and the idle gap after each kernel (rocprofv3 --kernel-trace reduced by
# tools/trace_gaps.py), lag pipeline, config 4's shard and config 2
export TMPDIR=/tmp
mkdir -p gpurun_out/r05k
for w in c4-shard c2; do
  rm -rf gpurun_out/r05k/$w
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05k/$w -- python3 bench.py --workload $w --no-cpu-baseline --no-kernel-events --no-odeint --steps 400 --warmup 40 > gpurun_out/r05k/$w.json 2> gpurun_out/r05k/$w.err
  python3 profiles/tools/trace_gaps.py gpurun_out/r05k/$w > gpurun_out/r05k/${w}_gaps.txt 2>&1
  find gpurun_out/r05k/$w -name "*kernel_trace.csv" -delete; find gpurun_out/r05k/$w -name "*.db" -delete
  head -12 gpurun_out/r05k/${w}_gaps.txt
done
