#!/bin/bash
# round 6, call a: the capture-lifetime child with and without the deferred release
set -o pipefail
O=gpurun_out/r06a; mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/env.txt 2>&1
echo "== with the deferred release" > $O/child.txt
timeout -k 10 300 python tests/_capture_lifetime_child.py >> $O/child.txt 2>&1; echo "exit code $?" >> $O/child.txt
echo "== --no-deferral (round 5's behaviour)" >> $O/child.txt
timeout -k 10 300 python tests/_capture_lifetime_child.py --no-deferral >> $O/child.txt 2>&1; echo "exit code $?" >> $O/child.txt
timeout -k 10 600 python -m pytest tests/test_gpu_capture_lifetime.py -m gpu -rA --tb=long -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/child.txt
cat $O/child.txt
