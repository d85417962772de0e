#!/bin/bash
# round 6, call e: odeint tail (trace reduction + host profile), free-running parity test, changed-kernel tests
set -o pipefail
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 profiles/tools/odeint_tail.py --calls 6 --plain > $O/trace.log 2>&1
echo "rocprof exit $?" >> $O/trace.log
timeout -k 10 100 python3 profiles/tools/odeint_tail.py --trace $O/trace > $O/odeint_trace.txt 2>&1
find $O/trace -name "*.csv" ! -name "*kernel_trace.csv" -delete; find $O/trace -name "*.db" -delete
cat $O/odeint_trace.txt
timeout -k 10 200 python3 profiles/tools/odeint_tail.py --host-profile > $O/host_profile.txt 2>&1
head -70 $O/host_profile.txt
timeout -k 10 900 python -m pytest tests/test_gpu_full_size_golden.py tests/test_gpu_kernels.py -m gpu -rA --tb=long -q -p no:cacheprovider -k "unmodified_bar or initial_step or control or pi_ or two_norms" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "^PASSED" $O/pytest.log | tail -40
