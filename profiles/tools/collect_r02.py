#!/usr/bin/env python3
"""Copy the summaries of profiles/tools/profile_r02.sh (its scratch output: gpurun_out/prof_r02/) into profiles/ as r02_*.

    python3 profiles/tools/collect_r02.py
"""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = os.path.join(ROOT, "gpurun_out", "prof_r02")
D = os.path.join(ROOT, "profiles")


def first_json(path):
    try:
        return json.load(open(path))
    except Exception:
        for ln in open(path):
            if ln.startswith("{"):
                return json.loads(ln)
    raise RuntimeError("no JSON in " + path)


def save(src, dst):
    json.dump(first_json(os.path.join(S, src)), open(os.path.join(D, dst), "w"), indent=1)


def main():
    for src, dst in [("bench_default.json", "r02_bench_default.json"), ("rk4.json", "r02_rk4.json"), ("bench_f64.json", "r02_bench_f64.json"),
                     ("force_dist.json", "r02_force_dist.json"), ("force_dist_p2p.json", "r02_force_dist_p2p.json"),
                     ("rehearsal_n2_allreduce.json", "r02_rehearsal_n2_allreduce.json"), ("rehearsal_n2_p2p.json", "r02_rehearsal_n2_p2p.json"),
                     ("c3_graph.json", "r02_c3_graph.json"), ("c3_eager.json", "r02_c3_eager.json"), ("c3_auto.json", "r02_c3_auto.json"), ("c1.json", "r02_c1.json"),
                     ("pmc_traffic.json", "r02_pmc_traffic.json"), ("pmc_traffic_rk4.json", "r02_pmc_traffic_rk4.json")]:
        save(src, dst)
    for p in ("graph", "auto", "sync", "lag"):
        save("c5_%s.json" % p, "r02_c5_%s.json" % p)
    rows = {}
    for b in (2048, 8192, 32768, 65536, 262144):
        rows[str(b)] = {}
        for p in ("auto", "sync", "lag", "graph"):
            j = first_json(os.path.join(S, "sweep_%d_%s.json" % (b, p)))
            rows[str(b)][p] = {"us_per_step": 1e3 * j["ms_per_step"], "states_per_s": j["value"], "resolved": j["config"]["pipeline"]}
    json.dump({"command": "python bench.py --batch B --pipeline P --no-cpu-baseline --steps 60 --warmup 20 (dim 128, fp32)", "rows": rows},
              open(os.path.join(D, "r02_pipeline_sweep.json"), "w"), indent=1)
    for src, dst in [("c5_graph_gaps.txt", "r02_c5_graph_gaps.txt"), ("ctrl_decomposition.txt", "r02_ctrl_decomposition.txt"),
                     ("graph_replay.txt", "r02_graph_replay.txt")]:
        shutil.copy(os.path.join(S, src), os.path.join(D, dst))
    for sub, dst in (("bench", "r02_bench_kernel_stats.csv"), ("rk4", "r02_rk4_kernel_stats.csv")):
        shutil.copy(glob.glob(os.path.join(S, sub, "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(D, dst))
    t = json.load(open(os.path.join(D, "r02_pmc_traffic.json")))
    c, cl, en = (t[k]["hbm_bytes_per_launch"] for k in ("combine", "combine_last_stage(+partial error)", "errnorm"))
    json.dump({"hbm_bytes_per_launch": (5 * c + cl) / 6,
               "source": "profiles/r02_pmc_traffic.json (5 plain stage launches + 1 last-stage launch per step)",
               "errnorm_hbm_bytes_per_launch": en}, open(os.path.join(D, "traffic_combine.json"), "w"), indent=1)
    j = json.load(open(os.path.join(D, "r02_bench_default.json")))
    print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"], {k: round(v["avg_us"], 2) for k, v in j["kernels"].items()})


if __name__ == "__main__":
    main()
