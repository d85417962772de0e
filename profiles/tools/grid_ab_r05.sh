# Round 5: grid of the few-stream stage launches (XDE_GRID_BLOCKS_FEW), alternating repetitions (see kernel_ab_r05.sh)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_grid_ab
REPS=${1:-8}
rm -rf $OUT; mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for W in c4-shard c2 rk4; do
    for g in 2048 1024 768 512; do
      d=$OUT/${W}_few${g}__$rep
      XDE_GRID_BLOCKS_FEW=$g rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload $W --no-cpu-baseline --no-kernel-events --no-odeint > $d.json 2> $d.err
      find $d -name "*kernel_trace.csv" -delete; find $d -name "*.db" -delete; find $d -name "*agent_info.csv" -delete
    done
  done
  echo "[$(date +%H:%M:%S)] rep $rep"
done
python3 profiles/tools/kernel_ab_r05.py $OUT | tee $OUT/ab.txt
