# Round 3, VERDICT r02 item 3: config 4's sizes on ONE GPU — (a) 65536 x 64, the per-GPU shard at N=8 (16 MiB operands, the whole
# working set Infinity-Cache resident), (b) 524288 x 64 on one rank, the N=1 point of the strong-scaling curve (128 MiB operands).
# Sweeps the cache policy (XDE_NT) and the launch grids at (a).  `gpurun -- 'bash profiles/tools/c4_sweep_r03.sh'`
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4_r03
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --steps 300 --warmup 30"
run() { # name, env..., -- args
  name=$1; shift
  envs=""
  while [ "$1" != "--" ]; do envs="$envs $1"; shift; done
  shift
  env $envs $B "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  echo "done $name"
}
run shard_default -- --batch 65536 --dim 64
for nt in 0 1 3 4 5 7; do run shard_nt$nt XDE_NT=$nt -- --batch 65536 --dim 64; done
run shard_nt7_all XDE_NT=7 XDE_NT_BYTES=4096 -- --batch 65536 --dim 64
for g in 256 512 1024 2048 4096; do run shard_normgrid$g XDE_NORM_GRID=$g -- --batch 65536 --dim 64; done
for g in 512 1024 4096; do run shard_grid$g XDE_GRID_BLOCKS=$g XDE_NORM_GRID=512 -- --batch 65536 --dim 64; done
for p in sync lag graph; do run shard_pipe_$p -- --batch 65536 --dim 64 --pipeline $p; done
run full_default -- --batch 524288 --dim 64
run full_nt0 XDE_NT=0 -- --batch 524288 --dim 64
run half_default -- --batch 262144 --dim 64
run quarter_default -- --batch 131072 --dim 64
python3 - <<'EOF'
import glob, json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/c4_r03"
rows = []
for f in sorted(glob.glob(out + "/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append((os.path.basename(f), "unreadable", str(e))); continue
    k = j.get("kernels", {})
    rows.append((os.path.basename(f)[:-5], "%.4g" % j["value"], "%.1f us/step" % (1e3 * j["ms_per_step"]),
                 "solver %.1f us" % (1e3 * j.get("solver_kernel_ms_per_step", 0)),
                 " ".join("%s=%.2fus/%sGB/s" % (n, r["avg_us"], ("%.0f" % r["algorithmic_GBps"]) if r.get("algorithmic_GBps") else "-") for n, r in k.items())))
with open(out + "/summary.txt", "w") as fh:
    for r in rows:
        fh.write(" | ".join(r) + "\n")
print(open(out + "/summary.txt").read())
EOF
