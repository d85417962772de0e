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
  ( export "${envs[@]}"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$label -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events --no-odeint > $OUT/$label.json 2> $OUT/$label.err )
  local f=$(find $OUT/$label -name "*kernel_stats.csv" | head -1)
  echo "## $label  (${envs[*]}; bench.py $*)  ms_per_step $(python3 -c "import json,sys; print(json.load(open('$OUT/$label.json')).get('ms_per_step'))" 2>/dev/null)" >> $OUT/ab.txt
  python3 - "$f" >> $OUT/ab.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "xde_" in r["Name"] and int(r["Calls"]) >= 100]
for r in rows:
    name = r["Name"].replace("void (anonymous namespace)::", "").replace("void xde::", "").split("(")[0]
    print("  %-78s calls %5s avg %8.2f us min %7.2f" % (name[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  find $OUT/$label -name "*kernel_trace.csv" -delete; find $OUT/$label -name "*.db" -delete
  echo "[$(date +%H:%M:%S)] $label"
}
W="--workload c4-shard"
one base_a XDE_COMBINE_LATE=1 XDE_COMBINE_PIPE=0 -- $W
one late0 XDE_COMBINE_LATE=0 XDE_COMBINE_PIPE=0 -- $W
one pipe1024 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=1024 -- $W
one pipe512 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=512 -- $W
one pipe2048 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=2048 -- $W
one base_b XDE_COMBINE_LATE=1 XDE_COMBINE_PIPE=0 -- $W
one pipe1024_b XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=1024 -- $W
one c2_pipe0 XDE_COMBINE_PIPE=0 --
one c2_pipe1024 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=1024 --
one c2_pipe2048 XDE_COMBINE_PIPE=1 XDE_COMBINE_PIPE_GRID=2048 --
for g in 128 256 512 1024; do one dde_$g XDE_LAG_GRID=$g -- --workload dde; done
cat $OUT/ab.txt
