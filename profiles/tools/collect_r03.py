#!/usr/bin/env python3
"""Copy the summaries of profiles/tools/profile_r03.sh (scratch output: gpurun_out/prof_r03/) into profiles/ as r03_*.

    python3 profiles/tools/collect_r03.py
"""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = os.path.join(ROOT, "gpurun_out", "prof_r03")
D = os.path.join(ROOT, "profiles")


def first_json(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise RuntimeError("no JSON in " + path)


def save(src, dst):
    json.dump(first_json(os.path.join(S, src)), open(os.path.join(D, dst), "w"), indent=1)


def xde_rows(stats_csv):
    """Our kernels' rows (and the framework GEMM func runs on) of a rocprofv3 --stats kernel summary."""
    rows = list(csv.DictReader(open(stats_csv)))
    keep = [r for r in rows if "xde_" in r["Name"] or (r["Name"].startswith("Cijk_") and int(r["Calls"]) >= 100)]
    return rows[0].keys(), keep


def main():
    names = ["bench_default", "c4-shard", "c4-n1", "dense", "dde", "self_launch_n2", "self_launch_n4", "force_dist_allreduce", "force_dist_rccl",
             "force_dist_c4shard_allreduce", "force_dist_c4shard_rccl", "host_floor", "host_floor_dist_allreduce", "host_floor_dist_rccl",
             "c5_graph", "c5_auto", "c3_auto", "c3_eager", "c1", "rk4"]
    for n in names:
        save(n + ".json", "r03_" + n.replace("-", "_") + ".json")
    for w in ("c4-shard", "c4-n1"):
        shutil.copy(os.path.join(S, w + "_pmc_traffic.json"), os.path.join(D, "r03_" + w.replace("-", "_") + "_pmc_traffic.json"))
    for sub in ("c4-shard_stats", "c4-n1_stats", "dense_stats", "dde_stats"):
        # (gpurun merges a call's output INTO gpurun_out/: an earlier call's file of another pid may still be there — take the newest)
        src = max(glob.glob(os.path.join(S, sub, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        fields, keep = xde_rows(src)
        with open(os.path.join(D, "r03_" + sub.replace("-", "_").replace("_stats", "_kernel_stats.csv")), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(fields))
            w.writeheader()
            w.writerows(keep)
    with open(os.path.join(D, "r03_self_launch_refused.txt"), "w") as fh:
        fh.write("$ python3 bench.py --gpus 2 --steps 20 --warmup 5      (one-GPU box, no XDE_BENCH_REHEARSAL)\n")
        fh.write(open(os.path.join(S, "self_launch_refused.err")).read())
        fh.write("stdout: {!r}\n".format(open(os.path.join(S, "self_launch_refused.out")).read()))
    # roofline.traffic of bench.py: PMC bytes per launch of the stage combine (mean of the step's 5 plain + 1 last-stage launches), per size
    t = json.load(open(os.path.join(D, "traffic_combine.json")))
    by_size = t.get("by_size", {"65536x128/f32": {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"],
                                                  "errnorm_hbm_bytes_per_launch": t.get("errnorm_hbm_bytes_per_launch"),
                                                  "source": t.get("source")}})
    for w, key in (("c4_shard", "65536x64/f32"), ("c4_n1", "524288x64/f32")):
        p = json.load(open(os.path.join(D, "r03_%s_pmc_traffic.json" % w)))
        c, cl, en = (p[k]["hbm_bytes_per_launch"] for k in ("combine", "combine_last_stage(+partial error)", "errnorm"))
        by_size[key] = {"hbm_bytes_per_launch": (5 * c + cl) / 6, "errnorm_hbm_bytes_per_launch": en,
                        "source": "profiles/r03_%s_pmc_traffic.json (5 plain stage launches + 1 last-stage launch per step)" % w}
    t["by_size"] = by_size
    json.dump(t, open(os.path.join(D, "traffic_combine.json"), "w"), indent=1)
    for n in ("bench_default", "c4_shard", "c4_n1"):
        j = json.load(open(os.path.join(D, "r03_%s.json" % n)))
        print(n, "%.4g states/s" % j["value"], "%.1f us/step" % (1e3 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"],
              {k: round(v["avg_us"], 2) for k, v in j["kernels"].items()})


if __name__ == "__main__":
    main()
