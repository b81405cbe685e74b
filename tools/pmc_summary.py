import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,cn), v in sorted(agg.items()):
        if "hmgpu" in k: print(d.split("/")[-1], k[:34], cn, "max %.1f MiB" % (max(v)/1024))
