#!/bin/bash
# round 4: the error-norm pass INSIDE the step (bench.py --workload c4-shard) under rocprofv3, old body vs xde_errnorm_pre_kernel, norm grids
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
out=gpurun_out/r04b
mkdir -p $out
for cfg in "0 512" "1 512" "1 1024" "1 2048"; do
  set -- $cfg
  d=$out/insitu_pre$1_g$2
  rm -rf $d
  XDE_ERRNORM_PRE=$1 XDE_NORM_GRID=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --no-odeint > $d.log 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "## XDE_ERRNORM_PRE=$1 XDE_NORM_GRID=$2"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "xde_" in r["Name"]:
        print("%-90s calls %5s avg %8.2f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
