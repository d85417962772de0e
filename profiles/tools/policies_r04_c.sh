# Round 4, after the captured interval solves went in: the whole GPU suite with the final library, then again with XDE_INTERVAL_GRAPH=0
# (`gpurun -- "bash profiles/tools/policies_r04_c.sh"`); results appended to profiles/r04_policies.txt
# (pytest writes straight into files under gpurun_out/: a run that prints nothing for 7 minutes is taken to be hung)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -q > gpurun_out/policies_r04_C_default.log 2>&1
XDE_INTERVAL_GRAPH=0 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py --deselect tests/test_gpu_demo.py > gpurun_out/policies_r04_C_off.log 2>&1
{
echo "== (defaults: the whole GPU suite, bench-contract and demo files included)"
grep -E "passed|failed|^FAILED|^E  " gpurun_out/policies_r04_C_default.log | cut -c1-250 | head -8
echo "== XDE_INTERVAL_GRAPH=0"
grep -E "passed|failed|^FAILED|^E  " gpurun_out/policies_r04_C_off.log | cut -c1-250 | head -8
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
} 2>&1 | tee gpurun_out/policies_r04_C.txt
