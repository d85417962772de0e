# Round-5 xde_lag_grad A/B after the finishing workgroup's loads went four lags at a time (run on the GPU box:
# `gpurun -- 'bash profiles/tools/lag_ab_r05.sh [reps]'`).  Variants ALTERNATE; each run is one `rocprofv3 --kernel-trace --stats` of
# `bench.py --workload dde`; kernel_ab_r05.py prints the median over the repetitions of every kernel's average.
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_lag_ab
REPS=${1:-5}
rm -rf $OUT; mkdir -p $OUT
one() {  # one <label> <rep> <env assignments...>
  local label=$1 rep=$2; shift 2
  local d=$OUT/${label}__$rep
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload dde --no-cpu-baseline --no-kernel-events --no-odeint > $d.json 2> $d.err )
  find $d -name "*kernel_trace.csv" -delete; find $d -name "*.db" -delete; find $d -name "*agent_info.csv" -delete
}
for rep in $(seq 1 $REPS); do
  for u in 4 8; do for g in 512 640 768 1024; do one dde_u${u}_g$g $rep XDE_LAG_UNROLL=$u XDE_LAG_GRID=$g; done; done
  echo "[$(date +%H:%M:%S)] rep $rep"
done
python3 profiles/tools/kernel_ab_r05.py $OUT | tee $OUT/ab.txt
