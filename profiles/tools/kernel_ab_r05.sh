# Round-5 kernel A/B (run on the GPU box: `gpurun -- 'bash profiles/tools/kernel_ab_r05.sh [reps]'`).  Variants ALTERNATE, `reps` times each
# (box drift within one call is +-0.4 us on an 11 us kernel: a single run per variant decides nothing); each run is one
# `rocprofv3 --kernel-trace --stats` of bench.py; kernel_ab_r05.py prints the MEDIAN over the repetitions of every kernel's average.
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_ab
REPS=${1:-8}
rm -rf $OUT; mkdir -p $OUT
one() {  # one <label> <rep> <env assignments...> -- <bench args...>
  local label=$1 rep=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local d=$OUT/${label}__$rep
  ( export "${envs[@]}"; rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events --no-odeint > $d.json 2> $d.err )
  find $d -name "*kernel_trace.csv" -delete; find $d -name "*.db" -delete; find $d -name "*agent_info.csv" -delete
}
for rep in $(seq 1 $REPS); do
  for W in c4-shard c2; do
    one ${W}_L0P0 $rep XDE_COMBINE_LATE=0 XDE_COMBINE_PIPE=0 -- --workload $W
    one ${W}_L1P0 $rep XDE_COMBINE_LATE=1 XDE_COMBINE_PIPE=0 -- --workload $W
    one ${W}_L1P1g1024 $rep XDE_COMBINE_LATE=1 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=1024 -- --workload $W
    one ${W}_L1P1g512 $rep XDE_COMBINE_LATE=1 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=512 -- --workload $W
    one ${W}_L1P0g1024all $rep XDE_COMBINE_LATE=1 XDE_COMBINE_PIPE=0 XDE_GRID_BLOCKS=1024 -- --workload $W
  done
  for g in 256 384 512 768; do one dde_g$g $rep XDE_LAG_GRID=$g -- --workload dde; done
  echo "[$(date +%H:%M:%S)] rep $rep"
done
python3 profiles/tools/kernel_ab_r05.py $OUT | tee $OUT/ab.txt
