# Round-2 profile collection (run on the GPU box: `gpurun -- 'bash profiles/tools/profile_r02.sh'`); summaries are copied
# from gpurun_out/prof_r02/ into profiles/ afterwards.  Counters are collected in their own passes (--pmc + --kernel-trace only).
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --no-cpu-baseline > $OUT/bench_rocprof.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rk4 -- python3 bench.py --workload rk4 > $OUT/rk4_rocprof.json 2> $OUT/rk4.err
python3 bench.py --workload rk4 > $OUT/rk4.json 2>> $OUT/rk4.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_write.err
python3 profiles/tools/pmc_summarise.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_rk4 -- python3 bench.py --workload rk4 --no-kernel-events --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_fetch_rk4.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_rk4 -- python3 bench.py --workload rk4 --no-kernel-events --steps 10 --warmup 3 > /dev/null 2> $OUT/pmc_write_rk4.err
python3 profiles/tools/pmc_summarise.py $OUT/pmc_fetch_rk4 $OUT/pmc_write_rk4 > $OUT/pmc_traffic_rk4.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/c5 -- python3 bench.py --workload c5 --pipeline graph > $OUT/c5_rocprof.json 2> $OUT/c5.err
python3 profiles/tools/trace_gaps.py $OUT/c5 > $OUT/c5_graph_gaps.txt 2>&1
for p in graph auto sync lag; do python3 bench.py --workload c5 --pipeline $p > $OUT/c5_$p.json 2>/dev/null; done
python3 bench.py --workload c3 --graph-func > $OUT/c3_graph.json 2>/dev/null
python3 bench.py --workload c3 --graph-func off > $OUT/c3_eager.json 2>/dev/null
python3 bench.py --workload c3 > $OUT/c3_auto.json 2>/dev/null
python3 bench.py --workload c1 > $OUT/c1.json 2>/dev/null
python3 bench.py --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2>/dev/null
XDE_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline > $OUT/force_dist.json 2> $OUT/force_dist.err
XDE_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --exchange p2p > $OUT/force_dist_p2p.json 2> $OUT/force_dist_p2p.err
for x in allreduce p2p; do XDE_BENCH_REHEARSAL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --exchange $x --steps 20 --warmup 5 > $OUT/rehearsal_n2_$x.json 2> $OUT/rehearsal_n2_$x.err; done
for b in 2048 8192 32768 65536 262144; do for p in auto sync lag graph; do python3 bench.py --batch $b --pipeline $p --no-cpu-baseline --steps 60 --warmup 20 > $OUT/sweep_${b}_$p.json 2>/dev/null; done; done
python3 profiles/tools/ctrl_bench.py > $OUT/ctrl_decomposition.txt 2>&1
python3 profiles/tools/graph_bench.py > $OUT/graph_replay.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
ls $OUT
