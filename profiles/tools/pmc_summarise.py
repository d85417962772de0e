#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE: they do not fit one pass on gfx950's TCC, and counters are
collected apart from any trace: `--pmc X --kernel-trace` only) into HBM bytes per launch for the solver's kernels.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python3 profiles/pmc_summarise.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

Units and the gfx950 correction are /opt/skills/guides/MI355X_MICROARCH.md's: both counters are in KiB; FETCH_SIZE reports
exactly HALF of the bytes of a wide coalesced streaming read on gfx950 (128-byte requests tallied at 64 B), so
hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  Infinity-Cache hits are counted in FETCH_SIZE (not excluded)."""
import csv
import glob
import json
import os
import re
import sys


def classify(name):
    if "xde_combine_pre_kernel" in name:
        return "combine_pre(partial sum in)"
    if "xde_combine_kernel" in name:
        # xde_combine_kernel<T, MODE, VEC, OUT2, ...> (round 5 added the cache policy and the operand-count split behind OUT2)
        m = re.search(r"xde_combine_kernel<([^>]*)>", name)
        args = [a.strip() for a in m.group(1).split(",")] if m else []
        if len(args) >= 4:
            mode, out2 = args[1], args[3] == "true"
            if mode == "1":
                return "combine_fuse(+partial final sum)" if out2 else "combine_fuse"
            if mode == "2":
                return "combine_wfuse"
            # (since round 4 "last stage" is also the stage that emits the next stage's partial sum: both write two arrays)
            return "combine_last_stage(+partial error)" if out2 else "combine"
        return "combine"
    for key, tag in (("xde_errnorm_control", "errnorm+control"), ("xde_errnorm", "errnorm"), ("xde_control", "control"),
                     ("xde_dense", "dense"), ("xde_commit", "commit"), ("xde_finalize", "finalize"), ("xde_p2p", "p2p_exchange")):
        if key in name:
            return tag
    if name.startswith("Cijk_") or "gemm" in name.lower():
        return "gemm(func)"
    return None


def read(dirname, counter):
    rows = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                tag = classify(r["Kernel_Name"])
                if tag is None:
                    continue
                rows.setdefault(tag, []).append(float(r["Counter_Value"]))
    return rows


def main():
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    out = {}
    for tag in sorted(set(fetch) | set(write)):
        f, w = fetch.get(tag, []), write.get(tag, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out[tag] = {
            "FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm,
            "hbm_bytes_per_launch": 2.0 * fm * 1024.0 + wm * 1024.0,
            "raw": {"FETCH_SIZE": {"dispatches": len(f), "min_KB": min(f) if f else None, "max_KB": max(f) if f else None},
                    "WRITE_SIZE": {"dispatches": len(w), "min_KB": min(w) if w else None, "max_KB": max(w) if w else None}},
        }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
