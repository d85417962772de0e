#!/bin/bash
# round 6, call g: odeint tail after the start-time evaluations moved ahead of the set-up; the whole GPU suite (full log)
set -o pipefail
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
{ timeout -k 10 200 python3 profiles/tools/odeint_tail.py --plain; timeout -k 10 200 python3 profiles/tools/odeint_tail.py; } > $O/odeint_tail.txt 2>&1
echo "tail exit $?" >> $O/odeint_tail.txt
grep "total_ms\|entry_host\|heuristic_host\|heuristic_gpu\|attempts_host" $O/odeint_tail.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06g/bench_default.json") if l.startswith("{")][0])
print({k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_blocks", "odeint_ms_T2", "odeint_ms_T11")})
print(j.get("odeint_T2"))
PY
timeout -k 10 850 python -m pytest tests -m gpu -rA --tb=long -q -p no:cacheprovider --durations=15 > $O/suite.log 2>&1
echo "suite exit $?" >> $O/suite.log
grep -v "^PASSED" $O/suite.log | tail -30
