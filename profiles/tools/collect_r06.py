#!/usr/bin/env python3
"""Copy the summaries of profiles/tools/profile_r06.sh (scratch output: gpurun_out/prof_r06/) into profiles/ as r06_*, and stamp the
recorded counter passes with the kernel sources they were taken on (bench.py reports `roofline.traffic` only for a matching stamp).

    python3 profiles/tools/collect_r06.py
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = os.path.join(ROOT, "gpurun_out", "prof_r06")
D = os.path.join(ROOT, "profiles")


def first_json(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise RuntimeError("no JSON in " + path)


def save(src, dst):
    p = os.path.join(S, src)
    if not os.path.exists(p):
        return False
    try:
        json.dump(first_json(p), open(os.path.join(D, dst), "w"), indent=1)
        return True
    except RuntimeError:
        return False


def xde_rows(stats_csv):
    """Our kernels' rows (and the framework GEMM func runs on, and RCCL's kernels) of a rocprofv3 --stats kernel summary."""
    rows = list(csv.DictReader(open(stats_csv)))
    keep = [r for r in rows if "xde_" in r["Name"] or (r["Name"].startswith("Cijk_") and int(r["Calls"]) >= 100) or "nccl" in r["Name"].lower()]
    return rows[0].keys(), keep


def main():
    from paddlexde_amd.csrc.build import kernel_stamp

    names = ["bench_default", "dense", "dde", "self_launch_n2", "self_launch_n3", "self_launch_n4", "host_floor", "host_floor_graph", "host_floor_dist_p2p_graph",
             "c5_graph", "c5_auto", "c3_auto", "c1", "rk4", "bench_f64"]
    names += ["force_dist_c4shard_" + x for x in ("p2p", "rccl", "allreduce")] + ["host_floor_dist_" + x for x in ("p2p", "rccl", "allreduce")]
    for n in names:
        save(n + ".json", "r06_" + n + ".json")
    save("c4-shard.json", "r06_c4_shard.json")
    save("c4-n1.json", "r06_c4_n1.json")
    for sub, dst in [("bench_stats", "bench_kernel_stats.csv"), ("c4-shard_stats", "c4_shard_kernel_stats.csv"), ("c4-n1_stats", "c4_n1_kernel_stats.csv"),
                     ("dist_stats", "force_dist_p2p_kernel_stats.csv"), ("dde_stats", "dde_kernel_stats.csv")]:
        # (every profiling run has a directory of its own, suffixed with its start time: the newest one counts)
        found = glob.glob(os.path.join(S, sub + "_*", "**", "*kernel_stats.csv"), recursive=True)
        if not found:
            continue
        fields, keep = xde_rows(max(found, key=os.path.getmtime))
        with open(os.path.join(D, "r06_" + dst), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(fields))
            w.writeheader()
            w.writerows(keep)
    # roofline.traffic of bench.py: PMC bytes per launch of the stage combine (mean of the step's 5 plain + 1 last-stage launches), per size
    tpath = os.path.join(D, "traffic_combine.json")
    try:
        by_size = {k: v for k, v in json.load(open(tpath)).get("by_size", {}).items() if v.get("round") == 6}
    except Exception:
        by_size = {}
    for name, key, dst in (("bench", "65536x128/f32", "r06_pmc_traffic.json"), ("c4-shard", "65536x64/f32", "r06_c4_shard_pmc_traffic.json"),
                           ("c4-n1", "524288x64/f32", "r06_c4_n1_pmc_traffic.json")):
        p = os.path.join(S, name + "_pmc_traffic.json")
        if not os.path.exists(p):
            continue
        pj = json.load(open(p))
        if "combine" not in pj:
            continue
        json.dump(pj, open(os.path.join(D, dst), "w"), indent=1)
        c, cl = (pj[k]["hbm_bytes_per_launch"] for k in ("combine", "combine_last_stage(+partial error)"))
        en = pj.get("errnorm", {}).get("hbm_bytes_per_launch")
        cp = pj.get("combine_pre(partial sum in)", {}).get("hbm_bytes_per_launch")
        # a Dopri5 step: 3 plain launches (stages 1-3), 2 two-output launches (stage 4 emits stage 5's partial sum, stage 6 the partial
        # error), 1 pre-summed launch (stage 5)
        mean = (3 * c + 2 * cl + cp) / 6 if cp else (5 * c + cl) / 6
        by_size[key] = {"hbm_bytes_per_launch": mean, "errnorm_hbm_bytes_per_launch": en, "round": 6,
                        "kernel_stamp": kernel_stamp("combine"), "errnorm_kernel_stamp": kernel_stamp("errnorm"),
                        "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; mean over the step's six stage launches)" % dst}
        # ... and rocprofv3's mean duration of the same six launches, from the kernel summary collected above (bench.py prints it beside
        # its own event timing: roofline.rocprofv3)
        stats = os.path.join(D, "r06_" + {"bench": "bench", "c4-shard": "c4_shard", "c4-n1": "c4_n1"}[name] + "_kernel_stats.csv")
        if os.path.exists(stats):
            tot = cnt = 0.0
            for r in csv.DictReader(open(stats)):
                n = r["Name"]
                stage = ("xde_combine_kernel<" in n and ", 0, " in n.split("xde_combine_kernel<")[1][:12]) or "xde_combine_pre_kernel<" in n
                if stage and int(r["Calls"]) >= 100:
                    tot += float(r["TotalDurationNs"])
                    cnt += int(r["Calls"])
            if cnt:
                by_size[key]["rocprofv3_avg_launch_us"] = tot / cnt / 1e3
                by_size[key]["rocprofv3_source"] = ("profiles/%s (rocprofv3 --kernel-trace --stats; calls-weighted mean over the step's six stage launches)"
                                                    % os.path.basename(stats))
    if by_size:
        json.dump({"by_size": by_size, "note": "written by profiles/tools/collect_r06.py; bench.py reports a figure only when kernel_stamp equals "
                   "csrc/build.py::kernel_stamp('combine') of the sources the library was built from"}, open(tpath, "w"), indent=1)
    # the A/B of the transports (alternating on one box)
    rows = []
    for rep in (1, 2, 3):
        for x in ("unsharded", "p2p", "rccl", "allreduce"):
            p = os.path.join(S, "ab_%s_%d.json" % (x, rep))
            if os.path.exists(p):
                try:
                    rows.append((rep, x, 1e3 * first_json(p)["ms_per_step"]))
                except RuntimeError:
                    rows.append((rep, x, float("nan")))
    if rows:
        with open(os.path.join(D, "r06_exchange_ab.txt"), "w") as fh:
            fh.write("# profiles/tools/profile_r06.sh (part `dist`): config 4's shard (65536 x 64) on ONE GPU, the sharded code path with one rank\n"
                     "# (XDE_BENCH_FORCE_DIST=1) per transport and the unsharded step, alternating three times on the same box; us per attempted step,\n"
                     "# no sampled events.  p2p = xde_p2p_rk_control (finalize + mailbox exchange + controller, ONE launch)\n")
            for rep, x, us in rows:
                fh.write("rep %d  %-10s %8.1f us/step\n" % (rep, x, us))
    p = os.path.join(S, "c3_timeline.txt")
    if os.path.exists(p):
        open(os.path.join(D, "r06_c3_timeline.txt"), "w").write(
            "# profiles/tools/c3_timeline.py on a rocprofv3 --kernel-trace of `python3 bench.py --workload c3` (round 6 library): the second, tuned\n"
            "# Dopri5 repetition (durations and the span are inflated by the tracer, the MIX is what counts)\n" + open(p).read())
    p = os.path.join(S, "watchdog.err")
    if os.path.exists(p):
        keep = [ln for ln in open(p).read().splitlines() if "bench.py" in ln or ln.startswith("  rank") or ln.startswith("rc=")]
        open(os.path.join(D, "r06_watchdog.txt"), "w").write(
            "$ XDE_BENCH_REHEARSAL=1 XDE_BENCH_TIMEOUT=60 XDE_BENCH_TEST_HANG=1 python3 bench.py --gpus 2 --steps 20 --warmup 5\n"
            "# (XDE_BENCH_TEST_HANG=r makes rank r stop at the start of its set-up: the job's clocks must end it and say where every rank was)\n"
            + "\n".join(keep) + "\nstdout: %r\n" % open(os.path.join(S, "watchdog.out")).read())
    for n in ("bench_default", "c4_shard", "c4_n1"):
        p = os.path.join(D, "r06_%s.json" % n)
        if os.path.exists(p):
            j = json.load(open(p))
            print(n, "%.4g states/s" % j["value"], "%.1f us/step" % (1e3 * j["ms_per_step"]), "frac %.3f" % j["roofline"]["frac"],
                  {k: round(v["avg_us"], 2) for k, v in j["kernels"].items()})


if __name__ == "__main__":
    main()
