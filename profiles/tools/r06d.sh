#!/bin/bash
# round 6, call d: whole-odeint() tail itemised; then the GPU suite on the split test files
set -o pipefail
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
{ timeout -k 10 200 python3 profiles/tools/odeint_tail.py --plain; timeout -k 10 200 python3 profiles/tools/odeint_tail.py; } > $O/odeint_tail.txt 2>&1
echo "tail exit $?" >> $O/odeint_tail.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 profiles/tools/odeint_tail.py --calls 6 --plain > $O/trace.log 2>&1
echo "rocprof exit $?" >> $O/trace.log
timeout -k 10 100 python3 profiles/tools/odeint_tail.py --trace $O/trace >> $O/odeint_tail.txt 2>&1
rm -rf $O/trace
cat $O/odeint_tail.txt
timeout -k 10 800 python -m pytest tests -m gpu -rA --tb=long -q -p no:cacheprovider --durations=15 > $O/suite.log 2>&1
echo "suite exit $?" >> $O/suite.log
tail -4 $O/suite.log
