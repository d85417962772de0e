# round 5, call e: the whole GPU suite on the final library, then the controller decomposition (XDE_CTRL_FLAGS variants), the RK4 line
mkdir -p gpurun_out/r05e
timeout -k 10 800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05e/suite.log 2>&1; echo rc=$? >> gpurun_out/r05e/suite.log; tail -3 gpurun_out/r05e/suite.log
for f in 7 0; do XDE_CTRL_FLAGS=$f python3 profiles/tools/ctrl_bench_r05.py c4 > gpurun_out/r05e/ctrl_c4_flags$f.txt 2>&1; done
python3 profiles/tools/ctrl_bench_r05.py c2 > gpurun_out/r05e/ctrl_c2_flags7.txt 2>&1
grep "^[A-F]" gpurun_out/r05e/ctrl_c4_flags7.txt
