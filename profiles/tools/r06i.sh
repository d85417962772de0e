#!/bin/bash
# round 6, call i: smoke(), the tests added after the last full suite run
set -o pipefail
O=gpurun_out/r06i; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -3 $O/smoke.txt
timeout -k 10 900 python -m pytest tests/test_gpu_full_size_golden.py tests/test_gpu_odeint.py tests/test_gpu_world8.py tests/test_gpu_capture_lifetime.py -m gpu -rA --tb=long -q -p no:cacheprovider -k "free_running or four_launches or world8 or capture or fully_free" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep "parity report\|passed\|failed\|exit" $O/pytest.log | tail -20
