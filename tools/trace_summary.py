"""per-kernel statistics of the BATCHED dispatches in a rocprofv3 --kernel-trace CSV (the bench stages its pictures with
one single-picture launch each before the timed region; everything before the first batched k_prep dispatch, the one
with Grid_Size_Z == batch, is left out here, so the averages can be compared with bench.py's hipEvent timings)
usage: python tools/trace_summary.py <kernel_trace.csv> [batch]"""
import collections
import csv
import sys

batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = collections.defaultdict(list)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("hmgpu::", "void hmgpu::"))]
t0 = min(int(r["Start_Timestamp"]) for r in rows if "k_prep" in r["Kernel_Name"] and int(r["Grid_Size_Z"]) == batch * int(r["Workgroup_Size_Z"]))
for r in rows:
    if int(r["Start_Timestamp"]) < t0:
        continue
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print('"Name","BatchedCalls","AverageNs","MinNs","MaxNs"')
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.1f,%d,%d' % (k, len(v), sum(v) / len(v), min(v), max(v)))
