"""Per-kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV (steady-state part of the run)."""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 2:]  # steady state
dur = collections.defaultdict(list); gap_after = collections.defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    key = n[:70]
    dur[key].append(e - s)
    if i + 1 < len(rows):
        gap_after[key].append(rows[i + 1][0] - e)
tot_d = sum(e - s for s, e, _ in rows); span = rows[-1][1] - rows[0][0]
print("kernels %d  span %.1f us  busy %.1f us (%.0f%%)" % (len(rows), span / 1e3, tot_d / 1e3, 100.0 * tot_d / span))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap_after[k]
    print("%6d x %7.2f us  gap-after %6.2f us   %s" % (len(d), sum(d) / len(d) / 1e3, (sum(g) / max(len(g), 1)) / 1e3, k))
