#!/usr/bin/env python3
"""Turns rocprofv3 --pmc CSVs (one pass per counter, as MI355X_MICROARCH.md prescribes: FETCH_SIZE
and WRITE_SIZE do not fit one pass) into profiles/<tag>_pmc_traffic.json:
    {kernel: {"launches": n, "fetch_kb": avg FETCH_SIZE, "write_kb": avg WRITE_SIZE,
              "hbm_bytes_per_launch": (2 * FETCH_SIZE + WRITE_SIZE) * 1024}}
The factor 2 is the gfx950 correction for FETCH_SIZE (it reports half of the bytes of a coalesced
streaming read; calibrated here on k_prepare, which reads exactly 8 B/read: 100e6 reads -> 390.6 MB
reported for 800 MB read).  WRITE_SIZE needs no correction (k_prepare writes 4 B/read: 391.1 MB).
usage: summarize_pmc.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json> [source note]"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main(src, out):
    fetch = load(os.path.join(src, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = load(os.path.join(src, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"), "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        res[k] = {"launches": len(fetch.get(k, write.get(k, []))), "fetch_kb": round(f, 1),
                  "write_kb": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    # per-solve total over the solver's own kernels (k_*; runtime copies and fills are not the solve's)
    per_solve = sum(v["hbm_bytes_per_launch"] for k, v in res.items() if k.startswith("k_"))
    res["_total"] = {"hbm_bytes_per_solve": per_solve}
    res["_source"] = (sys.argv[3] if len(sys.argv) > 3 else
                      "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass per counter (profiles/collect.sh)")
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")
    print(f"{'solver kernels, per solve':28s} {per_solve / 1e6:10.1f} MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
