# Round-5 kernel A/B at config 4's shard size and the D3STN-sized delay gradient (run on the GPU box:
# `gpurun -- 'bash profiles/tools/kernel_ab_r05.sh'`).  Each variant is one `rocprofv3 --kernel-trace --stats` run of bench.py; the per-kernel
# averages of this library's kernels are printed side by side into gpurun_out/r05_ab/ab.txt.
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_ab
mkdir -p $OUT
: > $OUT/ab.txt
one() {  # one <label> <env assignments...> -- <bench args...>
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  rm -rf $OUT/$label
  env "${envs[@]}" true  # (validates the assignments)
  ( export "${envs[@]}" XDE_AB=1; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$label -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events --no-odeint > $OUT/$label.json 2> $OUT/$label.err )
  local f=$(find $OUT/$label -name "*kernel_stats.csv" | head -1)
  echo "## $label  (${envs[*]}; bench.py $*)" >> $OUT/ab.txt
  python3 - "$f" >> $OUT/ab.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "xde_" in r["Name"]]
tot = 0.0
for r in rows:
    name = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    print("  %-78s calls %5s avg %8.2f us min %7.2f" % (name[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  find $OUT/$label -name "*kernel_trace.csv" -delete; find $OUT/$label -name "*.db" -delete
  echo "[$(date +%H:%M:%S)] $label"
}
for rep in 1 2; do
  one late1_$rep XDE_COMBINE_LATE=1 -- --workload c4-shard
  one late0_$rep XDE_COMBINE_LATE=0 -- --workload c4-shard
done
one c2_late1 XDE_COMBINE_LATE=1 --
one c2_late0 XDE_COMBINE_LATE=0 --
one grid1024 XDE_GRID_BLOCKS=1024 -- --workload c4-shard
one dde_1024 XDE_LAG_GRID=1024 -- --workload dde
one dde_512 XDE_LAG_GRID=512 -- --workload dde
one dde_2048 XDE_LAG_GRID=2048 -- --workload dde
one rk4 XDE_AB=1 -- --workload rk4
cat $OUT/ab.txt
