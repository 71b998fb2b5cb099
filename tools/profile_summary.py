#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small tracked files under profiles/.

    python tools/profile_summary.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>]

Writes profiles/<tag>_kernel_stats.csv (the --stats table as rocprofv3 emitted it),
profiles/<tag>_pmc_traffic.json (per-kernel HBM bytes per launch from FETCH_SIZE/WRITE_SIZE,
corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950, both x1024) and refreshes
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """kernel name without namespace and argument list; rocprofv3 leaves some names mangled (its demangler gives up on _Float16
    parameters): _ZN12_GLOBAL__N_1<len><name>[I(Li<n>E)+E]... -> name<n, ...>"""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1))
        rest = name[m.end():]
        ident, rest = rest[:n], rest[n:]
        t = re.match(r"I((?:Li\d+E|Lb[01]E)+)E", rest)
        if t:
            args = re.findall(r"L[ib](\d+)E", t.group(1))
            return ident + "<" + ", ".join(args) + ">"
        return ident
    m = re.search(r"([A-Za-z_0-9]+)(<[^>]*>)?\(", name.replace("(anonymous namespace)::", ""))
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def agg(d, ctr):
    out = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr:
                k = short(r["Kernel_Name"])
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    for path in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        f, w = agg(sys.argv[3], "FETCH_SIZE"), agg(sys.argv[4], "WRITE_SIZE")
        res = {}
        for k in sorted(set(f) | set(w)):
            fn, fs = f.get(k, [0, 0.0])
            wn, ws = w.get(k, [0, 0.0])
            fetch = 2.0 * 1024.0 * fs / max(fn, 1)   # gfx950: FETCH_SIZE reads half of a wide stream
            write = 1024.0 * ws / max(wn, 1)
            res[k] = {"launches_fetch_pass": fn, "launches_write_pass": wn,
                      "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                      "hbm_bytes_per_launch": fetch + write}
        res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KB units x1024; "
                        "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B). Averages are over all "
                        "launches of a kernel in one bench run (shapes differ between launches).")
        json.dump(res, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)
        json.dump(res, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
        for k, v in res.items():
            if isinstance(v, dict):
                print(f"{k:40s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
