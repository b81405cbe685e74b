"""summarise rocprofv3 --pmc CSVs: per kernel and counter, the value of the largest (= batched) dispatch.
usage: python tools/pmc_summary.py [--json out.json] <dir> [<dir> ...]
--json additionally writes the per-launch HBM traffic of every kernel, priced as MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE tallies 128-B read requests at 64 B)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


from pmc_summary_digest import csrc_digest

BENCH_NAME = {"k_itx": "itx", "k_mc_luma<false, false>": "mc_luma", "k_mc_chroma<false, false>": "mc_chroma", "k_mc_luma<false, false, true>": "mc_luma",
              "k_mc_chroma<false, false, true>": "mc_chroma", "k_mc_luma<false, false, false>": "mc_luma", "k_mc_chroma<false, false, false>": "mc_chroma", "k_deblock<0>": "deblock_ver",
              "k_deblock<1>": "deblock_hor", "k_sao": "sao", "k_prep": "prep", "k_prep<1>": "prep", "k_extend": "extend_border",
              "k_filter_fused<false>": "filter_fused", "k_intra": "intra"}
args = sys.argv[1:]
json_out = None
if "--json" in args:
    i = args.index("--json")
    json_out = args[i + 1]
    del args[i:i + 2]
rows = collections.defaultdict(dict)
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hmgpu::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, cn), v in agg.items():
            rows[k][cn] = max(v)
names = sorted({c for k in rows for c in rows[k]})
for k in sorted(rows):
    if k.startswith("k_"):
        print(k)
        for c in names:
            if c in rows[k]:
                print("    %-32s %16.0f" % (c, rows[k][c]))
if json_out:
    out = {}
    for k, v in rows.items():
        if k in BENCH_NAME and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[BENCH_NAME[k]] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                                  "traffic_bytes": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 --warmup 1 "
                         "--profile-steps 1 --no-cpu-baseline`; per launch = largest dispatch (batch of 16 pictures); "
                         "traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md",
               "csrc_sha16": csrc_digest(), "kernels": out}, open(json_out, "w"), indent=1)
