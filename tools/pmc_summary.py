"""summarise rocprofv3 --pmc CSVs: per kernel and counter, the value of the largest (= batched) dispatch"""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(dict)
for d in sys.argv[1:]:
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
