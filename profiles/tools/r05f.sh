# round 5, call f: checksummed mirror publish (XDE_CTRL_FLAGS bit 8) — controller decomposition 15 vs 7, the GPU suite under both, the RK4 A/B
mkdir -p gpurun_out/r05f
for f in 15 7 15 7; do XDE_CTRL_FLAGS=$f python3 profiles/tools/ctrl_bench_r05.py c4 > gpurun_out/r05f/ctrl_c4_flags${f}_$RANDOM.txt 2>&1; done
for f in gpurun_out/r05f/ctrl_c4_flags*.txt; do echo "== $f"; grep "^[A-F]" $f | head -8; done
timeout -k 10 800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05f/suite.log 2>&1; echo rc=$? >> gpurun_out/r05f/suite.log; tail -3 gpurun_out/r05f/suite.log
XDE_CTRL_FLAGS=7 timeout -k 10 300 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not bench and not sharded_gloo and not full_size and not demo" > gpurun_out/r05f/suite_flags7.log 2>&1; echo rc=$? >> gpurun_out/r05f/suite_flags7.log; tail -3 gpurun_out/r05f/suite_flags7.log
for rep in 1 2 3 4 5; do for m in on off; do python3 bench.py --workload rk4 --rk4-presum $m > gpurun_out/r05f/rk4_${m}_$rep.json 2>/dev/null; done; done
python3 - <<'PY'
import json, statistics
for m in ("on", "off"):
    rows = [json.load(open("gpurun_out/r05f/rk4_%s_%d.json" % (m, r))) for r in range(1, 6)]
    print("rk4 presum", m, "states/s median %.4g" % statistics.median(r["value"] for r in rows), "ms/step median %.4f" % statistics.median(r["ms_per_step"] for r in rows),
          "fuse avg %.2f us" % statistics.median(r["kernels"]["combine_fuse"]["avg_us"] for r in rows),
          "final %.2f us" % statistics.median(r["kernels"]["combine_wfuse"]["avg_us"] for r in rows),
          "kernels/step %.1f us" % statistics.median(1e3 * r["solver_kernel_ms_per_step"] for r in rows))
PY
