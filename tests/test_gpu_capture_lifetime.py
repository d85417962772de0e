"""Captured graphs must never be released while a stream capture is open (gpurun_out/r05f/suite.log)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cached_captures_that_die_inside_a_recording_do_not_abort_the_process():
    """Round 5's abort made deterministic, in a child process (tests/_capture_lifetime_child.py): two modules with cached captures —
    one inside a reference cycle, one not — lose their last reference INSIDE the stream capture of a third module's dynamics, whose
    `forward` also calls `gc.collect()` explicitly.  With the deferred release (utils/graphed.py: `CapturedGraph.__del__` ->
    `release_when_idle`) the graphs and their pools are destroyed when the recording has ended; the child ends with the captured
    gradients bit-equal to the eager route's and reports how many graphs were parked (>= 4)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_capture_lifetime_child.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, "exit code {}\n--- stdout\n{}\n--- stderr\n{}".format(r.returncode, r.stdout, r.stderr[-6000:])
    assert r.stdout.strip().splitlines()[-1].startswith("OK deferred="), r.stdout
