#!/bin/bash
# round 6, call b: capture-lifetime test (fixed count), world-8 in-process rehearsal, bench-contract file
set -o pipefail
O=gpurun_out/r06b; mkdir -p $O
timeout -k 10 300 python tests/_capture_lifetime_child.py > $O/child.txt 2>&1; echo "exit code $?" >> $O/child.txt
timeout -k 10 300 python tests/_world8_child.py exchange > $O/w8_exchange.txt 2>&1; rc=$?; echo "exit code $rc" >> $O/w8_exchange.txt
if [ $rc -eq 0 ]; then
  timeout -k 10 600 python tests/_world8_child.py solve sync > $O/w8_sync.txt 2>&1; echo "exit code $?" >> $O/w8_sync.txt
  timeout -k 10 600 python tests/_world8_child.py solve lag > $O/w8_lag.txt 2>&1; echo "exit code $?" >> $O/w8_lag.txt
fi
timeout -k 10 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_capture_lifetime.py -m gpu -rA --tb=long -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/child.txt $O/w8_*.txt; tail -15 $O/pytest.log
