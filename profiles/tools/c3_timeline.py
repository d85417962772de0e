"""Where config 3's backward spends its GPU time: kernel counts / durations / idle gaps of the last pass of `bench.py --workload c3`
(dopri5 forward+backward, then rk4) from a rocprofv3 --kernel-trace CSV, grouped into OURS (xde_*), COPIES (copyBuffer / fill) and the
FRAMEWORK's kernels (func forward, autograd, GEMMs)."""
import collections
import csv
import glob
import sys

path = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the dopri5 backward of the second repetition: the longest stretch between two launches of the forward's dense/commit is hard to find
# generically, so report the whole second half of the trace (rep 2 of dopri5 + both reps of rk4 are similar in mix)
ec = [i for i, r in enumerate(rows) if "xde_errnorm_control_single" in r[2]]  # one per attempted Dopri5 step: 2 repetitions of fwd + bwd
half = rows[ec[len(ec) // 2]: ec[-1] + 40] if len(ec) >= 4 else rows[len(rows) // 2:]  # = the second (tuned) Dopri5 repetition


def group(n):
    if "xde_" in n:
        return "ours:" + n.split("xde_")[1].split("<")[0].split("(")[0]
    if "copyBuffer" in n or "fillBuffer" in n or "FillFunctor" in n:
        return "copy/fill"
    if n.startswith("Cijk_"):
        return "framework:gemm"
    return "framework:other"


cnt, dur = collections.Counter(), collections.Counter()
for s, e, n in half:
    g = group(n)
    cnt[g] += 1
    dur[g] += e - s
span = half[-1][1] - half[0][0]
busy = sum(dur.values())
print("kernels %d, span %.2f ms, busy %.2f ms (%.0f%%), idle %.2f ms" % (len(half), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
for g in sorted(cnt, key=lambda g: -dur[g]):
    print("%-28s %7d launches  %8.2f ms  avg %6.2f us" % (g, cnt[g], dur[g] / 1e6, dur[g] / cnt[g] / 1e3))
gaps = sorted((half[i + 1][0] - half[i][1]) for i in range(len(half) - 1))
print("gap between consecutive kernels: median %.2f us, p90 %.2f us, max %.1f us, sum of gaps > 20 us: %.2f ms" % (
    gaps[len(gaps) // 2] / 1e3, gaps[int(0.9 * len(gaps))] / 1e3, gaps[-1] / 1e3, sum(g for g in gaps if g > 20000) / 1e6))
