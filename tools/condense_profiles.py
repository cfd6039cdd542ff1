#!/usr/bin/env python3
"""Condense gpurun_out/profiles_raw/ (tools/make_profiles.sh) into the tracked profiles/ directory:
kernel-stats CSVs, a per-kernel PMC table and profiles/pmc_traffic.json (HBM bytes per launch of the
streaming kernels; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads
on gfx950, WRITE_SIZE is taken as is; both counters are in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(REPO, "gpurun_out", "profiles_raw")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"
# on the GPU box (tools/make_profiles.sh) the condensed files go to gpurun_out/profiles_out (merged back by gpurun,
# the raw traces are deleted there); copy them into the tracked profiles/ directory afterwards
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles")


def first(pattern):
    f = glob.glob(os.path.join(RAW, pattern))
    return f[0] if f else None


def kernel_means(tag):
    f = first(f"{tag}/*/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not k.startswith("evc::") and "evc::" not in k:
            continue
        agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


os.makedirs(OUT, exist_ok=True)
for tag in ("stats_default", "stats_sym8_b32s1", "stats_pack2_b32s1", "stats_sym8_md", "stats_pack2_md",
            "stats_zundel100_b32s1", "stats_zundel100_md", "stats_h2ovtz_b4s1", "stats_h2ovtz_b32s1", "stats_h10_md"):
    f = first(f"{tag}/*/*kernel_stats.csv")
    if f:
        rows = [r for r in csv.DictReader(open(f)) if "evc::" in r["Name"]]
        with open(os.path.join(OUT, f"{ROUND}_kernel_stats_{tag[6:]}.csv"), "w", newline="") as fo:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
traffic = {}
table = []
for lay in ("sym8", "pack2"):
    for regime, key in (("b32", "batch32"), ("md", "batch1")):
        fe, wr = kernel_means(f"pmc_fetch_{lay}_{regime}"), kernel_means(f"pmc_write_{lay}_{regime}")
        for k in sorted(set(fe) | set(wr)):
            fetch_kib = fe.get(k, {}).get("FETCH_SIZE", 0.0)
            write_kib = wr.get(k, {}).get("WRITE_SIZE", 0.0)
            hbm = (2.0 * fetch_kib + write_kib) * 1024.0
            table.append((lay, regime, k, fetch_kib, write_kib, hbm))
            short = "k5" if "gemv_rows" in k and "reduce" not in k else "k8" if "gemv_cols" in k else None
            if short:
                traffic[f"H30/{lay}/{key}/{short}"] = {
                    "kernel": k, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
                    "hbm_bytes_per_launch": hbm,
                    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}
# provenance: bench.py only quotes these numbers while the streaming-kernel sources are the ones profiled
import hashlib
_repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in ("gemv_mfma.hip", "gemv_stream.hip", "gemv_lds.hip"):
    _h.update(open(os.path.join(_repo, "evcont_amd", "csrc", _f), "rb").read())
traffic["_source_sha256"] = _h.hexdigest()
json.dump(traffic, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(OUT, f"{ROUND}_pmc_hbm_traffic.csv"), "w") as fo:
    fo.write("layout,regime,kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_per_launch\n")
    for r in table:
        fo.write(",".join(str(x) for x in r) + "\n")
sq = kernel_means("pmc_sq_sym8_b32")
if sq:
    cols = sorted({c for d in sq.values() for c in d})
    with open(os.path.join(OUT, f"{ROUND}_pmc_sq_sym8_batch32.csv"), "w") as fo:
        fo.write("kernel," + ",".join(cols) + "\n")
        for k, d in sq.items():
            fo.write(k + "," + ",".join(f"{d.get(c, 0):.0f}" for c in cols) + "\n")
lds = kernel_means("pmc_lds_sym8_b32")   # LDS counters of the same run (a pass of its own)
if lds:
    cols = sorted({c for d in lds.values() for c in d})
    with open(os.path.join(OUT, f"{ROUND}_pmc_lds_sym8_batch32.csv"), "w") as fo:
        fo.write("kernel," + ",".join(cols) + "\n")
        for k, d in lds.items():
            fo.write(k + "," + ",".join(f"{d.get(c, 0):.0f}" for c in cols) + "\n")
for tag, name in (("pmc_sq_h2ovtz_b4", "pmc_sq_h2ovtz_batch4"), ("pmc_lds_h2ovtz_b4", "pmc_lds_h2ovtz_batch4")):
    m = kernel_means(tag)   # the 64-wide kernels of csrc/pair64.hip (cc-pVTZ water shape, four geometries per batch)
    if m:
        cols = sorted({c for d in m.values() for c in d})
        with open(os.path.join(OUT, f"{ROUND}_{name}.csv"), "w") as fo:
            fo.write("kernel," + ",".join(cols) + "\n")
            for k, d in m.items():
                fo.write(k + "," + ",".join(f"{d.get(c, 0):.0f}" for c in cols) + "\n")
print(json.dumps(traffic, indent=1))
