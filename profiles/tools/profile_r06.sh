# Round-6 profile collection (run on the GPU box: `gpurun -- 'bash profiles/tools/profile_r06.sh [part ...]'`); summaries are copied from
# gpurun_out/prof_r06/ into profiles/ by tools/collect_r06.py.  Counters are collected in their own passes (--pmc + --kernel-trace only);
# rocprofv3 is always given `python3 bench.py ...` directly after `--`.  Parts: headline c4 dist side (default: all).
# Every rocprofv3 run writes into a directory of its own (suffix = this run's start time); collect_r06.py takes the newest.
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/prof_r06
RUN=$(date +%s)
mkdir -p $OUT
say() { echo "[$(date +%H:%M:%S)] $*"; }
PARTS="${*:-headline c4 dist side}"
want() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
pmc() {  # pmc <name> <bench args...>: the two counter passes of one workload -> $OUT/<name>_pmc_traffic.json
  local name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_pmc_fetch_$RUN -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events --no-odeint --steps 10 --warmup 3 > /dev/null 2> $OUT/${name}_pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_pmc_write_$RUN -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events --no-odeint --steps 10 --warmup 3 > /dev/null 2> $OUT/${name}_pmc_write.err
  python3 profiles/tools/pmc_summarise.py $OUT/${name}_pmc_fetch_$RUN $OUT/${name}_pmc_write_$RUN > $OUT/${name}_pmc_traffic.json
  say "$name pmc"
}
stats() {  # stats <name> <bench args...>: rocprofv3 --kernel-trace --stats of one workload -> $OUT/<name>_stats_<run>/
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats_$RUN -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events --no-odeint > $OUT/${name}_rocprof.json 2> $OUT/${name}_rocprof.err
  say "$name stats"
}
if want headline; then
  # the driver's N=1 line (with the whole-odeint() calls and both CPU baselines), its kernel summary, its counter passes
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say default
  stats bench
  pmc bench
fi
if want c4; then
  for w in c4-shard c4-n1; do
    python3 bench.py --workload $w --no-cpu-baseline > $OUT/${w}.json 2> $OUT/${w}.err; say "$w events"
    stats $w --workload $w
    pmc $w --workload $w
  done
fi
if want dist; then
  # the sharded code path with ONE rank: the three transports and the unsharded step, alternating, at config 4's shard; the host's enqueue
  # floor (tiny state) under lag and under graph replay; the self-launched N-rank invocation rehearsed on this one GPU
  for rep in 1 2 3; do
    python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events > $OUT/ab_unsharded_$rep.json 2>/dev/null
    for x in p2p rccl allreduce; do
      XDE_BENCH_FORCE_DIST=1 python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --exchange $x > $OUT/ab_${x}_$rep.json 2> $OUT/ab_${x}_$rep.err
    done
  done; say ab
  for x in p2p rccl allreduce; do
    XDE_BENCH_FORCE_DIST=1 python3 bench.py --workload c4-shard --no-cpu-baseline --exchange $x > $OUT/force_dist_c4shard_$x.json 2> $OUT/force_dist_$x.err
    XDE_BENCH_FORCE_DIST=1 python3 bench.py --batch 256 --dim 64 --pipeline lag --no-kernel-events --no-cpu-baseline --steps 2000 --warmup 200 --exchange $x > $OUT/host_floor_dist_$x.json 2>> $OUT/force_dist_$x.err
  done
  XDE_BENCH_FORCE_DIST=1 python3 bench.py --batch 256 --dim 64 --pipeline graph --no-kernel-events --no-cpu-baseline --steps 2000 --warmup 200 --exchange p2p > $OUT/host_floor_dist_p2p_graph.json 2>> $OUT/force_dist_p2p.err
  python3 bench.py --batch 256 --dim 64 --pipeline lag --no-kernel-events --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/host_floor.json 2>/dev/null
  python3 bench.py --batch 256 --dim 64 --pipeline graph --no-kernel-events --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/host_floor_graph.json 2>/dev/null; say floors
  XDE_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dist_stats_$RUN -- python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --no-odeint --exchange p2p > /dev/null 2> $OUT/dist_stats.err; say "dist stats"
  XDE_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/self_launch_n2.json 2> $OUT/self_launch_n2.err; echo "rc=$?" >> $OUT/self_launch_n2.err; say n2
  XDE_BENCH_REHEARSAL=1 python3 bench.py --gpus 3 --steps 20 --warmup 5 --batch 65536 > $OUT/self_launch_n3.json 2> $OUT/self_launch_n3.err; echo "rc=$?" >> $OUT/self_launch_n3.err; say n3
  # the watchdog: a job whose ranks never finish is stopped by the parent, with the stage every rank was in
  XDE_BENCH_REHEARSAL=1 XDE_BENCH_TIMEOUT=60 XDE_BENCH_TEST_HANG=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/watchdog.out 2> $OUT/watchdog.err; echo "rc=$?" >> $OUT/watchdog.err; say watchdog
fi
if want side; then
  for w in dense dde; do
    python3 bench.py --workload $w > $OUT/${w}.json 2> $OUT/${w}.err; say "$w events"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dde_stats_$RUN -- python3 bench.py --workload dde > /dev/null 2> $OUT/dde_stats.err; say "dde stats"
  for p in graph auto; do python3 bench.py --workload c5 --pipeline $p > $OUT/c5_$p.json 2>/dev/null; done; say c5
  python3 bench.py --workload c3 > $OUT/c3_auto.json 2>/dev/null; say c3
  rocprofv3 --kernel-trace --output-format csv -d $OUT/c3_trace_$RUN -- python3 bench.py --workload c3 > /dev/null 2>&1
  python3 profiles/tools/c3_timeline.py $OUT/c3_trace_$RUN > $OUT/c3_timeline.txt 2>&1; say "c3 timeline"
  python3 bench.py --workload c1 > $OUT/c1.json 2>/dev/null; say c1
  python3 bench.py --workload rk4 > $OUT/rk4.json 2>/dev/null; say rk4
  python3 bench.py --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2>/dev/null; say f64
fi
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls $OUT
