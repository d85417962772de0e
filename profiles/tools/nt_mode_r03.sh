# Which operands of the stage combines to stream (non-temporal loads) at config 2's size (32 MiB operands) — alternating on one box.
mkdir -p gpurun_out/r03d
for rep in 1 2 3; do for m in lastuse old oldk all none; do
  XDE_STAGE_NT_MODE=$m python3 bench.py --no-cpu-baseline --no-kernel-events --steps 400 --warmup 40 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 rep $rep $m', round(1e3*j['ms_per_step'],1))"
done; done
for rep in 1 2; do for m in lastuse old oldk none; do
  XDE_STAGE_NT_MODE=$m python3 bench.py --workload c4-shard --no-cpu-baseline --no-kernel-events --steps 400 --warmup 40 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard rep $rep $m', round(1e3*j['ms_per_step'],1))"
done; done
