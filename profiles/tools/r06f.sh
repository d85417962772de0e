#!/bin/bash
# round 6, call f: odeint tail after the host trims; free-running parity; the tests of everything touched
set -o pipefail
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
{ timeout -k 10 200 python3 profiles/tools/odeint_tail.py --plain; timeout -k 10 200 python3 profiles/tools/odeint_tail.py; } > $O/odeint_tail.txt 2>&1
echo "tail exit $?" >> $O/odeint_tail.txt
cat $O/odeint_tail.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06f/bench_default.json") if l.startswith("{")][0])
print({k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_blocks", "odeint_ms_T2", "odeint_ms_T11")})
print(j.get("odeint_T2"))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_full_size_golden.py tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -rA --tb=long -q -p no:cacheprovider -k "unmodified_bar or free_running or initial_step or control or pi_ or two_norms or four_launches or lag or speculat or peek or interval" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "^PASSED" $O/pytest.log | tail -60
