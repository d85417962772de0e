"""Median over repetitions of every kernel's average duration, per variant (see kernel_ab_r05.sh)."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

out = sys.argv[1]
data = collections.defaultdict(lambda: collections.defaultdict(list))  # variant -> kernel -> [avg us per rep]
steps = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(out, "*__*"))):
    if not os.path.isdir(d):
        continue
    variant = os.path.basename(d).rsplit("__", 1)[0]
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        if "xde_" in r["Name"] and int(r["Calls"]) >= 100:
            name = r["Name"].replace("void (anonymous namespace)::", "").replace("void xde::", "").split("(")[0]
            data[variant][name].append(float(r["AverageNs"]) / 1e3)
    try:
        j = json.load(open(d + ".json"))
        if j.get("ms_per_step"):
            steps[variant].append(1e3 * j["ms_per_step"])
    except Exception:
        pass
for variant in sorted(data):
    n = max(len(v) for v in data[variant].values())
    extra = ""
    if steps[variant]:
        extra = "  step median %.1f us (min %.1f)" % (statistics.median(steps[variant]), min(steps[variant]))
    print("## %s  (%d runs)%s" % (variant, n, extra))
    tot = 0.0
    for name, v in sorted(data[variant].items(), key=lambda kv: -statistics.median(kv[1])):
        print("  %-72s median %7.2f us  min %7.2f  max %7.2f" % (name[:72], statistics.median(v), min(v), max(v)))
