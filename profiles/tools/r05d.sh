mkdir -p gpurun_out/r05d
timeout -k 10 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05d/suite.log 2>&1; echo rc=$? >> gpurun_out/r05d/suite.log; tail -3 gpurun_out/r05d/suite.log
for f in 7 3 1 0; do XDE_CTRL_FLAGS=$f python3 profiles/tools/ctrl_bench_r05.py c4 > gpurun_out/r05d/ctrl_c4_flags$f.txt 2>&1; done
python3 profiles/tools/ctrl_bench_r05.py c2 > gpurun_out/r05d/ctrl_c2_flags7.txt 2>&1
tail -20 gpurun_out/r05d/ctrl_c4_flags7.txt
for rep in 1 2 3; do python3 bench.py --workload rk4 > gpurun_out/r05d/rk4_$rep.json 2> gpurun_out/r05d/rk4_$rep.err; done
python3 -c "
import json
for r in (1,2,3):
    j=json.load(open('gpurun_out/r05d/rk4_%d.json'%r)); print('rk4', j.get('value'), j.get('ms_per_step'), {k:(v['avg_us']) for k,v in j.get('kernels',{}).items()})
"
