# round 5, call g: the whole GPU suite on the final library (checksummed mirror publish = default; cyclic GC held off during captures)
mkdir -p gpurun_out/r05g
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 > gpurun_out/r05g/suite.log 2>&1; echo rc=$? >> gpurun_out/r05g/suite.log; tail -4 gpurun_out/r05g/suite.log
