# round 5, call h: occupancy cap (unused dynamic LDS) for the everything-streamed stage launches at config 4's whole problem (128 MiB operands)
export TMPDIR=/tmp
OUT=gpurun_out/r05h; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3 4; do
  for lds in 1 24576 36864 49152 65536; do
    d=$OUT/lds${lds}__$rep
    XDE_BIG_LDS=$lds rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload c4-n1 --no-cpu-baseline --no-kernel-events --no-odeint > $d.json 2> $d.err
    find $d -name "*kernel_trace.csv" -delete; find $d -name "*.db" -delete
  done
  for g in 1024 1536; do
    d=$OUT/grid${g}__$rep
    XDE_GRID_BLOCKS=$g rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload c4-n1 --no-cpu-baseline --no-kernel-events --no-odeint > $d.json 2> $d.err
    find $d -name "*kernel_trace.csv" -delete; find $d -name "*.db" -delete
  done
  echo "[$(date +%H:%M:%S)] rep $rep"
done
python3 profiles/tools/kernel_ab_r05.py $OUT | tee $OUT/ab.txt
