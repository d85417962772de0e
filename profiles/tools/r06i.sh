#!/bin/bash
# round 6, call i: smoke(), the tests added after the last full suite run, the lines a forced collection per capture had inflated
set -o pipefail
O=gpurun_out/r06i; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -3 $O/smoke.txt
for p in graph auto; do timeout -k 10 200 python3 bench.py --workload c5 --pipeline $p > $O/c5_$p.json 2>/dev/null; done
timeout -k 10 200 python3 bench.py --workload c3 > $O/c3_auto.json 2>/dev/null
timeout -k 10 200 python3 bench.py --workload c1 > $O/c1.json 2>/dev/null
python - <<'PY'
import json
for w in ("c5_graph","c5_auto","c3_auto","c1"):
    j=json.loads([l for l in open("gpurun_out/r06i/%s.json"%w) if l.startswith("{")][0])
    r=j.get("results",{})
    print(w, {k:(round(v.get("us_per_attempted_step",0),1) if "us_per_attempted_step" in v else {kk:round(vv,5) for kk,vv in v.items() if kk.endswith("_s")}) for k,v in r.items()} or {k:v for k,v in j.items() if isinstance(v,(int,float))})
PY
timeout -k 10 900 python -m pytest tests/test_gpu_full_size_golden.py tests/test_gpu_odeint.py tests/test_gpu_world8.py tests/test_gpu_capture_lifetime.py -m gpu -rA --tb=long -q -p no:cacheprovider -k "free_running or four_launches or world8 or capture or fully_free" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep "parity report\|passed\|failed\|exit" $O/pytest.log | tail -20
