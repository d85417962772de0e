#!/bin/bash
# round 6, call c: the folded initial step of large states (tests + the bench line's whole-odeint() figures)
set -o pipefail
O=gpurun_out/r06c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -rA --tb=long -q -k "initial_step or two_norms or four_launches or interval_solves or adjoint_sizes or larger_state" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06c/bench_default.json") if l.startswith("{")][0])
print({k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_blocks", "odeint_ms_T2", "odeint_ms_T11")})
print(j.get("odeint_T2"))
PY
