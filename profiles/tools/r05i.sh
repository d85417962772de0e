# round 5, call i: the driver's path (build + smoke + default bench) and every side workload after bench_side.py was split off
mkdir -p gpurun_out/r05i
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r05i/smoke.log 2>&1; echo rc=$? >> gpurun_out/r05i/smoke.log; tail -2 gpurun_out/r05i/smoke.log
python bench.py > gpurun_out/r05i/bench_default.json 2> gpurun_out/r05i/bench_default.err; echo "bench rc=$?"
for w in c1 c3 c5 rk4 dense dde; do python bench.py --workload $w > gpurun_out/r05i/$w.json 2> gpurun_out/r05i/$w.err; echo "$w rc=$? $(head -c 150 gpurun_out/r05i/$w.json)"; done
timeout -k 10 500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_demo.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05i/contract.log 2>&1; echo rc=$? >> gpurun_out/r05i/contract.log; tail -3 gpurun_out/r05i/contract.log
timeout -k 10 200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "time_gradients_equal or vjp_hook" > gpurun_out/r05i/new.log 2>&1; echo rc=$? >> gpurun_out/r05i/new.log; tail -3 gpurun_out/r05i/new.log
