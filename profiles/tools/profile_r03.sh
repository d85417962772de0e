# Round-3 profile collection (run on the GPU box: `gpurun -- 'bash profiles/tools/profile_r03.sh'`); summaries are copied from
# gpurun_out/prof_r03/ into profiles/ afterwards.  Counters are collected in their own passes (--pmc + --kernel-trace only);
# rocprofv3 is always given `python3 bench.py ...` directly after `--`.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
say() { echo "[$(date +%H:%M:%S)] $*"; }
# 1. the driver's N=1 line
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say default
# 2. config 4's sizes on one GPU: HIP events, rocprofv3 kernel stats, PMC traffic
for w in c4-shard c4-n1; do
  python3 bench.py --workload $w --no-cpu-baseline > $OUT/${w}.json 2> $OUT/${w}.err; say $w events
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_stats -- python3 bench.py --workload $w --no-cpu-baseline --no-kernel-events > $OUT/${w}_rocprof.json 2> $OUT/${w}_rocprof.err; say $w stats
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${w}_pmc_fetch -- python3 bench.py --workload $w --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 > /dev/null 2> $OUT/${w}_pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${w}_pmc_write -- python3 bench.py --workload $w --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 > /dev/null 2> $OUT/${w}_pmc_write.err
  python3 profiles/tools/pmc_summarise.py $OUT/${w}_pmc_fetch $OUT/${w}_pmc_write > $OUT/${w}_pmc_traffic.json; say $w pmc
done
# 3. the two kernels that had no roofline line: dense output, history gather
for w in dense dde; do
  python3 bench.py --workload $w > $OUT/${w}.json 2> $OUT/${w}.err; say $w events
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_stats -- python3 bench.py --workload $w > $OUT/${w}_rocprof.json 2> $OUT/${w}_rocprof.err; say $w stats
done
# 4. `python bench.py --gpus 2` with NO launcher (rehearsal: both ranks on this GPU over gloo) and the refusal without the flag
XDE_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/self_launch_n2.json 2> $OUT/self_launch_n2.err; echo "rc=$?" >> $OUT/self_launch_n2.err; say n2
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/self_launch_refused.out 2> $OUT/self_launch_refused.err; echo "rc=$?" >> $OUT/self_launch_refused.err; say refused
XDE_BENCH_REHEARSAL=1 python3 bench.py --gpus 4 --steps 20 --warmup 5 > $OUT/self_launch_n4.json 2> $OUT/self_launch_n4.err; echo "rc=$?" >> $OUT/self_launch_n4.err; say n4
# the sharded code path with ONE rank over the nccl backend: what finalize -> all-reduce -> controller-on-sums costs per step, with
# the all-reduce through torch.distributed and as an in-stream ncclAllReduce (RcclExchange); and the host's enqueue floor (tiny state)
for x in allreduce rccl; do
  XDE_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --exchange $x > $OUT/force_dist_$x.json 2> $OUT/force_dist_$x.err
  XDE_BENCH_FORCE_DIST=1 python3 bench.py --workload c4-shard --no-cpu-baseline --exchange $x > $OUT/force_dist_c4shard_$x.json 2>> $OUT/force_dist_$x.err
  XDE_BENCH_FORCE_DIST=1 python3 bench.py --batch 256 --dim 64 --pipeline lag --no-kernel-events --no-cpu-baseline --steps 2000 --warmup 200 --exchange $x > $OUT/host_floor_dist_$x.json 2>> $OUT/force_dist_$x.err
done
python3 bench.py --batch 256 --dim 64 --pipeline lag --no-kernel-events --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/host_floor.json 2>/dev/null; say force_dist
# 5. side workloads
for p in graph auto; do python3 bench.py --workload c5 --pipeline $p > $OUT/c5_$p.json 2>/dev/null; done; say c5
python3 bench.py --workload c3 > $OUT/c3_auto.json 2>/dev/null; say c3
python3 bench.py --workload c1 > $OUT/c1.json 2>/dev/null; say c1
python3 bench.py --workload rk4 > $OUT/rk4.json 2>/dev/null; say rk4
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls $OUT
