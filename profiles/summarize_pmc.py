#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (one pass per counter, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not
fit one pass) -> per-kernel HBM bytes.

  summarize_pmc.py calibrate <dir> <out.json>
      <dir>/calib_FETCH_SIZE, <dir>/calib_WRITE_SIZE: passes over lab/pmc_calib, whose kernels move a KNOWN number of
      bytes with the access widths the solver uses.  Writes bytes per counter unit (the counters report KiB):
      {"fetch": {"read_2": f, "read_4": f, ...}, "write": {...}}.  The guide's gfx950 correction -- FETCH_SIZE
      reports half of a 16-B-per-lane streaming read -- is checked here for 2-, 4- and 8-byte lanes too.
  summarize_pmc.py summarize <dir> <prefix> <out.json> <calibration.json> [source note]
      <dir>/<prefix>pmc_FETCH_SIZE, <dir>/<prefix>pmc_WRITE_SIZE: passes over bench.py.  Per kernel:
      {"launches", "fetch_kb", "write_kb", "hbm_bytes_per_launch"} with
      hbm_bytes = fetch_kb * 1024 * F + write_kb * 1024 * W, F and W the calibrated factors of the kernel's access
      widths (KERNEL_WIDTHS below; a kernel that mixes widths takes the factor of the width that carries most of
      its bytes -- the calibration shows how far apart they are)."""
import collections
import csv
import json
import os
import re
import sys

CALIB_BYTES = 512 << 20
# which calibration kernel stands for a solver kernel's reads / writes (dominant stream by bytes)
KERNEL_WIDTHS = {
    "k_prepare": ("read_4", "write_16"),          # dword loads of starts and ends; 16-byte table runs, 8-byte mask clears
    "k_range_partition": ("read_4", "write_4"),   # dword loads; u16 + u32 stores (4 of every 6 bytes are dwords)
    "k_range_offsets": ("read_8", "write_4"),     # 8-byte loads of four u16 keys; dword stores of bucket offsets
    "k_rank_mark": ("rw_2_4", "atomic_or64"),     # u16 + u32 record streams; 64-bit atomic ORs
    "k_pm_prepare_sort": ("read_4", "write_4"),   # dword loads of starts and ends; dword stores of two 16-bit records
    "k_pm_offsets": ("read_8", "write_4"),        # 8-byte loads of four u16 keys; dword stores of bucket offsets
    "k_pm_rank_mark": ("read_2", "atomic_or64"),  # two u16 record streams; 64-bit atomic ORs
    "k_pm_descr": ("read_4", "write_4"),          # dword loads of the two tables; dword stores of descriptors
    "k_pm_range_table": ("read_4", "write_4"),
    "k_pm_walk": ("read_2", "atomic_or64"),       # two u16 record streams (+ dword descriptors, quota gathers); 64-bit atomic ORs, 64 per instruction
    "k_pm_settle": ("read_2", "atomic_or64"),
    "k_sweep_pack": ("read_4", "write_16"),
    "k_sweep_uniform_ev": ("read_16", "write_4"),
    "k_sweep_expand": ("read_4", "write_4"),
}


def short(name):
    m = re.search(r"(k_\w+|calib_\w+<[^>]*>|calib_\w+)", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def calib_key(kernel):
    if kernel.startswith("calib_rw_2_4"):
        return "rw_2_4"
    if kernel.startswith("calib_atomic_or64"):
        return "atomic_or64"
    m = re.match(r"calib_(read|write)<(.*)>", kernel)
    if not m:
        return None
    width = {"unsigned short": 2, "unsigned int": 4}.get(m.group(2))
    if width is None:
        width = 8 if "2" in m.group(2) else 16   # HIP_vector_type<unsigned int, 2u> / 4u
    return f"{m.group(1)}_{width}"


def calibrate(src, out):
    fetch = load(os.path.join(src, "calib_FETCH_SIZE", "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = load(os.path.join(src, "calib_WRITE_SIZE", "pmc_counter_collection.csv"), "WRITE_SIZE")
    res = {"fetch": {}, "write": {}, "raw_kb": {}}
    for k in sorted(set(fetch) | set(write)):
        key = calib_key(k)
        if key is None:
            continue
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        res["raw_kb"][key] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1)}
        moved = CALIB_BYTES // 6 * 6 if key == "rw_2_4" else CALIB_BYTES
        if key.startswith("read") or key == "rw_2_4":
            res["fetch"][key] = round(moved / (f * 1024), 4) if f else None
        if key.startswith("write"):
            res["write"][key] = round(moved / (w * 1024), 4) if w else None
        if key == "atomic_or64":
            # 4 Mi atomics: bytes each one moves to and from memory, as the counters see them
            res["atomic_or64"] = {"fetch_bytes_per_atomic": round(f * 1024 / (4 << 20), 2),
                                  "write_bytes_per_atomic": round(w * 1024 / (4 << 20), 2)}
    res["_note"] = ("bytes moved per KiB the counter reports, by access width (lab/pmc_calib.hip: 512 MiB per kernel, "
                    "coalesced); 1.0 = the counter is exact, 2.0 = it reports half")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def summarize(src, prefix, out, calib_path, note):
    fetch = load(os.path.join(src, prefix + "pmc_FETCH_SIZE", "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = load(os.path.join(src, prefix + "pmc_WRITE_SIZE", "pmc_counter_collection.csv"), "WRITE_SIZE")
    calib = json.load(open(calib_path)) if calib_path and os.path.exists(calib_path) else {"fetch": {}, "write": {}}
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [])), 1)
        rkey, wkey = KERNEL_WIDTHS.get(k, ("read_4", "write_4"))
        ff = calib["fetch"].get(rkey) or 2.0
        wf = (calib["write"].get(wkey) or 1.0) if wkey != "atomic_or64" else 1.0
        res[k] = {"launches": len(fetch.get(k, write.get(k, []))), "fetch_kb": round(f, 1), "write_kb": round(w, 1),
                  "fetch_factor": ff, "write_factor": wf, "hbm_bytes_per_launch": int((ff * f + wf * w) * 1024)}
    # per-solve total over the solver's own kernels (k_*; runtime copies and fills are not the solve's)
    per_solve = sum(v["hbm_bytes_per_launch"] for k, v in res.items() if k.startswith("k_"))
    res["_total"] = {"hbm_bytes_per_solve": per_solve}
    res["_source"] = note or "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass per counter (profiles/collect.sh)"
    res["_calibration"] = {"fetch": calib.get("fetch"), "write": calib.get("write"), "atomic_or64": calib.get("atomic_or64")}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  (fetch x{v['fetch_factor']}, write x{v['write_factor']})")
    print(f"{'solver kernels, per solve':28s} {per_solve / 1e6:10.1f} MB")


if __name__ == "__main__":
    if sys.argv[1] == "calibrate":
        calibrate(sys.argv[2], sys.argv[3])
    else:
        summarize(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None,
                  sys.argv[6] if len(sys.argv) > 6 else None)
