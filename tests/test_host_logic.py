"""Host logic on CPU: the cases of tests/_e2e_cases.py with the numpy test double of the kernel backend
(tests/_cpu_double.py).  Exercises the product's Python drivers (solver loops, operand plans, pipelines,
tuple flattening, adjoint) without a GPU; the HIP kernels themselves are covered by the `-m gpu` suite."""
import pytest

from ._dde_cases import *  # noqa: F401,F403
from ._e2e_cases import *  # noqa: F401,F403
from ._kernel_oracle_cases import *  # noqa: F401,F403
from ._replay_cases import *  # noqa: F401,F403


@pytest.fixture
def dev(cpu_double):
    return "cpu"


def test_bench_refuses_more_ranks_than_gpus_without_touching_a_gpu():
    """`python bench.py --gpus N` with no launcher starts its own ranks; on a box with fewer GPUs it must refuse (exit code != 0,
    no JSON line) rather than print an N=1 number — and a launcher's WORLD_SIZE that contradicts --gpus is refused too."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "XDE_BENCH_REHEARSAL")}
    import torch

    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3"], capture_output=True, text=True,
                           timeout=120, env=env)
        assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout
    env["WORLD_SIZE"], env["RANK"] = "2", "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "{" not in r.stdout


def test_bench_stage_watchdog_ends_a_rank_that_outlives_its_stage(tmp_path):
    """bench.py's per-rank stage clock (no GPU needed): a stage that outlives its limit ends the PROCESS with exit code 70, a line on stderr
    naming rank and stage, and a status file the launching parent can read; `on_expire(stage)` hooks may choose another NON-ZERO exit code
    (the extra measurements of the N > 1 line print the main line, marked with the stage, and leave with 75) — a hook that returns 0 does not
    turn a hang into a success."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, {root!r}); import bench\n"
            "wd = bench.Watchdog(3)\nwd.stage('rendezvous', 100)\nwd.stage('communicator', 0.5{hook})\ntime.sleep(30)\n")
    env = dict(os.environ, XDE_BENCH_STATUS_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code.format(root=root, hook="")], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 70 and "bench.py[rank 3]: stage 'communicator'" in r.stderr and "rendezvous" in r.stderr
    st = json.load(open(tmp_path / "rank3.json"))
    assert st["stage"] == "communicator" and st["note"] == "stage limit exceeded" and st["history"][0][0] == "rendezvous"
    sys.path.insert(0, root)
    import bench

    assert "rank 3: in stage 'communicator'" in bench._stage_report(str(tmp_path), 4) and "rank 0: never reported" in bench._stage_report(str(tmp_path), 4)
    r = subprocess.run([sys.executable, "-c", code.format(root=root, hook=", on_expire=lambda stage: 75 if stage == 'communicator' else 1")],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 75 and "over its limit" in r.stderr
    r = subprocess.run([sys.executable, "-c", code.format(root=root, hook=", on_expire=lambda stage: 0")], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 70 and "over its limit" in r.stderr


def test_bench_recorded_measurements_are_tied_to_the_kernel_sources(monkeypatch):
    """`roofline.traffic` (PMC counter passes) and `roofline.rocprofv3` (the committed kernel summary's mean stage-launch duration) are
    RECORDED measurements: bench.py reports them only while the stamp of the stage-combine kernel's sources equals the stamp the record
    was taken with, and says why when it does not."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from paddlexde_amd.csrc import build

    traffic, src = bench.pmc_traffic(65536, 128, "f32")
    if isinstance(src, dict) and "stale" in src:  # (the kernel was edited after the committed passes: the line says so and reports nothing)
        assert traffic is None
        pytest.skip("profiles/traffic_combine.json is older than csrc/xde_combine.hip: re-take the counter passes (profiles/tools/profile_r05.sh)")
    assert traffic and abs(traffic / (31 * 65536 * 128 * 4 / 6) - 1) < 0.01  # (PMC bytes = the 31 N 4 B a step's six stage launches move)
    assert src["kernel_stamp"] == src["current_kernel_stamp"] == build.kernel_stamp("combine")
    assert 20 < src["rocprofv3_avg_launch_us"] < 35 and "kernel_stats.csv" in src["rocprofv3_file"]
    monkeypatch.setattr(build, "kernel_stamp", lambda which: "0" * 12)  # (the kernel changed after the record was taken)
    traffic, src = bench.pmc_traffic(65536, 128, "f32")
    assert traffic is None and "stale" in src
    assert bench.pmc_traffic(123, 7, "f32")[0] is None


def test_garbage_collector_is_held_off_while_a_capture_records():
    """Round 5: an automatic cyclic collection that starts INSIDE a stream capture can reap a dropped module's cached captures (the
    per-module cache is keyed weakly) and release their graphs / memory pools in the middle of the recording — the process aborted
    under `weakref.remove` inside a captured func (gpurun_out/r05f/suite.log).  Every recording now runs with the collector off and
    restores it afterwards — also when the body raises, and without switching it ON for a caller who had it off."""
    import gc

    from paddlexde_amd.utils.graphed import _capture_without_gc

    class Ctx:
        def __init__(self):
            self.events = []

        def __enter__(self):
            self.events.append(("enter", gc.isenabled()))  # (torch.cuda.graph collects on entry: the collector is still on then)
            return "graph"

        def __exit__(self, *exc):
            self.events.append(("exit", gc.isenabled()))
            return False

    assert gc.isenabled()
    c = Ctx()
    with _capture_without_gc(c) as g:
        assert g == "graph" and not gc.isenabled()
    assert gc.isenabled() and c.events == [("enter", True), ("exit", False)]
    with pytest.raises(RuntimeError):
        with _capture_without_gc(Ctx()):
            raise RuntimeError("func cannot be captured")
    assert gc.isenabled()
    gc.disable()
    try:
        with _capture_without_gc(Ctx()):
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
