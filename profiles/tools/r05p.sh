# round 5, call p: the GPU suite on the round's last tree with the parity report and the durations, the default bench line, a 10x soak of
# the randomised sweeps
mkdir -p gpurun_out/r05p
rm -f gpurun_out/r05p/parity_report.jsonl
XDE_PARITY_REPORT=gpurun_out/r05p/parity_report.jsonl timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=40 > gpurun_out/r05p/suite.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/r05p/suite.log; tail -3 gpurun_out/r05p/suite.log
[ $rc -eq 0 ] || exit $rc
python3 bench.py > gpurun_out/r05p/bench_default.json 2> gpurun_out/r05p/bench_default.err; echo "bench rc=$?"
XDE_SWEEP_SCALE=10 timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "randomised" > gpurun_out/r05p/soak.log 2>&1; echo rc=$? >> gpurun_out/r05p/soak.log; tail -3 gpurun_out/r05p/soak.log
