# Round 5: the two-rank rehearsal of `bench.py --gpus 2` on one card, N times, WITH the settle phase (hundreds of mailbox exchanges between two
# processes time-slicing the card) — looking for the one red run of gpurun_out/r05w.  Keeps every run's line and stderr.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_soak
rm -rf $OUT; mkdir -p $OUT
N=${1:-8}
for i in $(seq 1 $N); do
  XDE_BENCH_REHEARSAL=1 XDE_BENCH_SETTLE_IN_REHEARSAL=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-n1 > $OUT/run$i.json 2> $OUT/run$i.err
  rc=$?
  python3 - $OUT/run$i.json $rc <<'PY'
import json, sys
try:
    j = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][0])
    rep = j["norm_exchange_report"]
    print("rc", sys.argv[2], "probe", rep["p2p_probe"]["ok"], "tried", rep["tried"], "settle", j["solver"]["settle_steps"], "ms/step %.3f" % j["ms_per_step"], "ab", sorted(j.get("exchange_ab", {})))
except Exception as e:
    print("rc", sys.argv[2], "no line:", type(e).__name__, e)
PY
done
