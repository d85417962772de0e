mkdir -p gpurun_out/r03b
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_odeint.py -m gpu -x -q -k "dense or taken_apart or step_at or manual_step or golden or lag_pipeline or graph" > gpurun_out/r03b/dense_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03b/dense_tests.log
for g in 512 1024 2048; do XDE_DENSE_GRID=$g python3 bench.py --workload dense > gpurun_out/r03b/dense_g$g.json 2>/dev/null; done
XDE_NT=0 python3 bench.py --workload dense > gpurun_out/r03b/dense_nt0.json 2>/dev/null
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03b/dense_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
    print(f, "avg_us=%.2f frac=%.3f T11=%.3f ms T2=%.3f ms"%(r["avg_launch_us"], r["frac"], j["solve_ms_T11"], j["solve_ms_T2"]))
PY
