# round 5, call l: lag vs graph pipeline at config 4's shard and config 2 on one rank (unsharded, and the sharded path over p2p), alternating
mkdir -p gpurun_out/r05l
for rep in 1 2 3; do
  for pl in lag graph; do
    python3 bench.py --workload c4-shard --pipeline $pl --no-cpu-baseline --no-kernel-events --steps 200 --warmup 40 > gpurun_out/r05l/shard_${pl}_$rep.json 2>/dev/null
    XDE_BENCH_FORCE_DIST=1 python3 bench.py --workload c4-shard --pipeline $pl --exchange p2p --no-cpu-baseline --no-kernel-events --steps 200 --warmup 40 > gpurun_out/r05l/shard_p2p_${pl}_$rep.json 2>/dev/null
    python3 bench.py --pipeline $pl --no-cpu-baseline --no-kernel-events --no-odeint --steps 200 --warmup 40 > gpurun_out/r05l/c2_${pl}_$rep.json 2>/dev/null
  done
done
python3 - <<'PY'
import json, glob, statistics
for w in ("shard", "shard_p2p", "c2"):
    for pl in ("lag", "graph"):
        v = []
        for r in (1, 2, 3):
            try:
                j = json.load(open("gpurun_out/r05l/%s_%s_%d.json" % (w, pl, r)))
                v.append([1e3 * b for b in j["ms_per_step_blocks"]])
            except Exception as e:
                v.append(["err", str(e)[:60]])
        print(w, pl, v)
PY
