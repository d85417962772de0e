# A/B of the norm transports on one rank (nccl group of one), config 4's shard, alternating on the same box.
mkdir -p gpurun_out/r03c
for rep in 1 2 3; do for x in allreduce rccl; do
  XDE_BENCH_FORCE_DIST=1 python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --steps 400 --warmup 40 --exchange $x 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep $x', round(1e3*j['ms_per_step'],1))"
done; done
python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --steps 400 --warmup 40 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unsharded', round(1e3*j['ms_per_step'],1))"
export TMPDIR=/tmp
for x in allreduce rccl; do
  XDE_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c/trace_$x -- python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --steps 100 --warmup 20 --exchange $x > /dev/null 2> gpurun_out/r03c/trace_$x.err
  f=$(find gpurun_out/r03c/trace_$x -name "*kernel_stats.csv" | head -1)
  echo "== $x"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if ("nccl" in n.lower() or "rccl" in n.lower() or "xde_finalize" in n or "xde_control" in n or "AllReduce" in n or "memcpy" in n.lower() or "copy" in n.lower()) and int(r["Calls"])>=50:
        print("%-90s calls=%s avg=%.2f us"%(n[:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
find gpurun_out/r03c -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out/r03c -name "*.db" -delete
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mall_bw profiles/tools/mall_bw.hip 2>/dev/null && /tmp/mall_bw > gpurun_out/r03c/mall_bw.txt; cat gpurun_out/r03c/mall_bw.txt
