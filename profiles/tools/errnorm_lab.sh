#!/bin/bash
# round 4: profiles/tools/errnorm_lab.hip under rocprofv3, grids 512 / 1024 / 2048 at 16 MiB per stream (config 4's shard) and 32 MiB (config 2)
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/lab
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gpurun_out/lab/errnorm_lab profiles/tools/errnorm_lab.hip
export TMPDIR=/tmp
for mib in ${LAB_MIB:-16 32}; do
  for grid in ${LAB_GRIDS:-512 1024 2048}; do
    d=gpurun_out/lab/m${mib}_g${grid}
    rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- gpurun_out/lab/errnorm_lab $mib $grid ${LAB_INSITU:-0} > $d.log 2>&1
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    echo "## ${mib} MiB per stream, grid ${grid}"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("k_") or "k_plain" in n or "k_pipe" in n or "k_bare" in n:
        if "k_fill" in n or "k_touch" in n: continue
        print("%-60s calls %4s avg %8.2f us  min %8.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
    elif "k_heat" in n or "k_write3" in n:
        print("%-60s calls %4s avg %8.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
