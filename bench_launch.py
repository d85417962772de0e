"""bench.py's process plumbing: the per-rank stage watchdog, the parent that starts one rank per GPU when `bench.py --gpus N` is run
without a launcher (and stops a job that never finishes), and the child-process probe of the peer-to-peer transport.  Nothing here
touches the GPU or is part of a timed region."""
import json
import os
import sys
import time

import torch

BENCH_PY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")


def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class Watchdog:
    """Per-rank stage clock.  `stage(name, seconds)` says what this rank is doing now and how long it may take; a daemon thread
    ends the process (os._exit, after saying on stderr which stage of which rank ran out) when a stage outlives its limit — a
    rendezvous that never completes, a communicator initialisation a peer never joins, a kernel that never returns.  Every
    stage change is also written to $XDE_BENCH_STATUS_DIR/rank<r>.json, which the launching parent reads when it has to kill the
    job.  `on_expire(stage) -> exit code` lets a stage say goodbye in its own way (the extra measurements print the main line, marked
    with the stage that ran out) — an expiry never ends in exit code 0.  XDE_BENCH_STAGE_SCALE multiplies every limit."""

    def __init__(self, rank):
        import threading

        self.rank = rank
        self.dir = os.environ.get("XDE_BENCH_STATUS_DIR")
        self.scale = float(os.environ.get("XDE_BENCH_STAGE_SCALE", "1"))
        self._lock = threading.Lock()
        self._stage, self._since, self._deadline, self._on_expire = None, None, None, None
        self._history = []
        threading.Thread(target=self._run, name="bench-watchdog", daemon=True).start()

    def _write(self, note=None):
        if not self.dir:
            return
        try:
            tmp = os.path.join(self.dir, "rank{}.json.tmp".format(self.rank))
            with open(tmp, "w") as fh:
                json.dump({"rank": self.rank, "pid": os.getpid(), "stage": self._stage, "since": self._since, "note": note,
                           "history": self._history}, fh)
            os.replace(tmp, os.path.join(self.dir, "rank{}.json".format(self.rank)))
        except Exception:
            pass

    def stage(self, name, seconds, on_expire=None):
        now = time.time()
        with self._lock:
            if self._stage is not None:
                self._history.append([self._stage, round(now - self._since, 3)])
            self._stage, self._since = name, now
            self._deadline = None if seconds is None else now + seconds * self.scale
            self._on_expire = on_expire
            self._write()

    def done(self):
        self.stage("done", None)

    def _run(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                late = self._deadline is not None and time.time() > self._deadline
                if late:
                    stage, since, hook = self._stage, self._since, self._on_expire
                    self._deadline = None
            if late:
                print("bench.py[rank {}]: stage '{}' has run for {:.0f} s, over its limit — giving up (stages so far: {})".format(
                    self.rank, stage, time.time() - since, self._history), file=sys.stderr, flush=True)
                self._write(note="stage limit exceeded")
                code = 70
                if hook is not None:
                    try:
                        code = hook(stage)
                    except Exception as e:
                        print("bench.py[rank {}]: {}".format(self.rank, e), file=sys.stderr, flush=True)
                os._exit(70 if not code else code)  # (a hang is never reported as success, whatever the hook returns)


def _stage_report(status_dir, n):
    """What the ranks of a job last said they were doing (the parent's diagnosis when it has to stop the job)."""
    rows = []
    for r in range(n):
        try:
            j = json.load(open(os.path.join(status_dir, "rank{}.json".format(r))))
            rows.append("  rank {}: in stage '{}' for {:.0f} s{}; before that: {}".format(
                r, j.get("stage"), time.time() - (j.get("since") or time.time()), " ({})".format(j["note"]) if j.get("note") else "",
                ", ".join("{} {:.1f}s".format(a, b) for a, b in j.get("history", [])) or "-"))
        except Exception:
            rows.append("  rank {}: never reported a stage (it did not get as far as bench.py's main)".format(r))
    return "\n".join(rows)


def group_members(pgid):
    """Live (non-zombie) processes whose process group or session is `pgid` — read from /proc, no signal is sent.  A zombie has
    released everything it held (its GPU context included) and only waits for its parent's `wait`."""
    out = []
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            with open("/proc/{}/stat".format(name)) as fh:
                stat = fh.read()
        except OSError:
            continue  # gone between the listing and the read
        # pid (comm) state ppid pgrp session ...: comm may hold spaces and parentheses, so split after the LAST ')'
        fields = stat[stat.rfind(")") + 2:].split()
        if len(fields) < 4 or fields[0] == "Z":
            continue
        if int(fields[2]) == pgid or int(fields[3]) == pgid:
            out.append(int(name))
    return out


def wait_for_group_exit(pgid, seconds):
    """Poll until no live process of process group / session `pgid` is left, `seconds` at most.  -> the pids still alive ([] = none)."""
    deadline = time.time() + seconds
    while True:
        left = group_members(pgid)
        if not left or time.time() > deadline:
            return left
        time.sleep(0.05)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) invoked WITHOUT a launcher (the reference's own recipe is one command too,
    example/D3STN/README.md:53-59): start one rank per GPU with `python -m torch.distributed.run` as a CHILD process, relay rank 0's
    JSON line, return non-zero if any rank fails or the line does not say `n_gpus == N`.  The parent never initialises the GPU
    and never re-execs itself.  On a box with fewer than N GPUs this refuses, unless XDE_BENCH_REHEARSAL=1 (all ranks share
    cuda:0, gloo carries the collectives: a functional rehearsal, not a scaling number).

    The parent is also the job's wall clock: the child runs in its own process group, and when XDE_BENCH_TIMEOUT seconds (default
    1500) pass without the job ending, that group is killed (TERM, then KILL), the stage every rank last reported is printed, and the
    exit code is 124.  Nothing is retried."""
    import signal
    import subprocess
    import tempfile

    n = args.gpus
    rehearsal = os.environ.get("XDE_BENCH_REHEARSAL", "0") == "1"
    have = torch.cuda.device_count()  # (does not initialise the GPU)
    if have < n and not rehearsal:
        print("bench.py: --gpus {} but this box has {} GPU(s); refusing to print a mislabelled line "
              "(XDE_BENCH_REHEARSAL=1 rehearses the N-rank invocation on one GPU over gloo)".format(n, have), file=sys.stderr)
        return 2
    if rehearsal and n > 6:
        # (the GPU pool's process guard ends a job with more than 6 processes on one card; the world-8 geometry is rehearsed in ONE
        # process instead: tests/test_gpu_world8.py — eight ranks as threads, real mailboxes, real kernels)
        print("bench.py: a rehearsal keeps to 6 ranks on one card", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    status_dir = tempfile.mkdtemp(prefix="xde_bench_status_")
    env["XDE_BENCH_STATUS_DIR"] = status_dir
    limit = float(os.environ.get("XDE_BENCH_TIMEOUT", "1500"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH_PY] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        print("bench.py: the {}-rank job is still running after {:.0f} s (XDE_BENCH_TIMEOUT): stopping it.  Last reported stages:\n{}".format(
            n, limit, _stage_report(status_dir, n)), file=sys.stderr, flush=True)
        for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
            try:
                os.killpg(proc.pid, sig)  # the child's own process group: the launcher and every rank it started, nothing else
            except ProcessLookupError:
                break
            try:
                proc.communicate(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        # `communicate` reaps the LAUNCHER only.  The ranks (and their probe children) are its children, not ours: killed, they may still
        # be tearing their GPU contexts down when the launcher is gone — whoever runs next on this card (the next test's probe
        # children, round 5's one red run) would meet that.  So: wait until nobody of that session is left before saying 124.
        left = wait_for_group_exit(proc.pid, 30.0)
        if left:
            print("bench.py: processes of the stopped job still alive after 30 s: {}".format(left), file=sys.stderr, flush=True)
        return 124
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0:
        print("bench.py: the {}-rank job exited with {}.  Last reported stages:\n{}".format(n, proc.returncode, _stage_report(status_dir, n)),
              file=sys.stderr)
        if len(lines) == 1 and "watchdog_expired" in lines[0]:
            print(lines[0], flush=True)  # the headline survived an extra measurement that hung: relayed, marked, and the exit code says so
        return proc.returncode or 1
    import shutil

    shutil.rmtree(status_dir, ignore_errors=True)
    if len(lines) != 1:
        print("bench.py: expected ONE JSON line from rank 0, got {}".format(len(lines)), file=sys.stderr)
        return 3
    try:
        got = json.loads(lines[0]).get("n_gpus")
    except Exception:
        got = None
    if got != n:
        print("bench.py: the line says n_gpus={} but --gpus {} was asked".format(got, n), file=sys.stderr)
        return 4
    print(lines[0], flush=True)
    return 0


def run_p2p_probe(dist, rank, world):
    """First contact of the peer-to-peer transport with this machine, made in CHILD processes (one per rank, started before this
    rank has touched its GPU): they form their own gloo group on a fresh port, map each other's mailboxes, push known vectors through
    the exchange and a short sharded solve through the fused controller launch.  A child that crashes, faults or hangs takes
    nothing of this job with it; the ranks then agree on the outcome.  -> {"ok": bool, "why": str, "seconds": float}"""
    import subprocess

    from paddlexde_amd.utils import exchange as X

    box = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    env = dict(os.environ)
    env["MASTER_PORT"] = str(box[0])
    env.pop("XDE_BENCH_STATUS_DIR", None)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # the children rendezvous on their own store (rank 0's child hosts it)
    t0 = time.perf_counter()
    why = None
    try:
        r = subprocess.run([sys.executable, BENCH_PY, "--probe-p2p", "--gpus", str(world)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=200)
        ok = r.returncode == 0
        if not ok:
            tail = (r.stderr or "").strip().splitlines()[-3:]
            why = "rank {}'s probe exited with {}: {}".format(rank, r.returncode, " | ".join(tail)[-400:])
    except subprocess.TimeoutExpired:
        ok, why = False, "rank {}'s probe did not finish in 200 s".format(rank)
    except Exception as e:
        ok, why = False, "rank {}: {}: {}".format(rank, type(e).__name__, e)
    all_ok = X.agree(ok)
    whys = [None] * world
    dist.all_gather_object(whys, why)
    return {"ok": all_ok, "why": "; ".join(w for w in whys if w) or None, "seconds": time.perf_counter() - t0}
