#!/bin/bash
# round 6, call j: a 5-rank one-card rehearsal of the self-launched bench (5 rank processes: the pool allows 6 on a card), and the randomised
# sweeps at 10x their default number of seeded blocks on the round's library (the fp32 controller's pow changed this round)
set -o pipefail
O=gpurun_out/r06j; mkdir -p $O
XDE_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 5 --steps 20 --warmup 5 --batch 65536 > $O/self_launch_n5.json 2> $O/self_launch_n5.err; echo "rc=$?" >> $O/self_launch_n5.err
tail -3 $O/self_launch_n5.err; head -c 600 $O/self_launch_n5.json; echo
XDE_SWEEP_SCALE=10 timeout -k 10 800 python -m pytest tests/test_gpu_odeint.py -m gpu -rA --tb=long -q -p no:cacheprovider -k "randomised" > $O/sweeps.log 2>&1
echo "sweeps exit $?" >> $O/sweeps.log
grep "passed\|failed\|exit" $O/sweeps.log | tail -5
