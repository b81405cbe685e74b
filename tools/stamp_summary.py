"""summary of /tmp/mc_stamps.bin written by an MC_STAMP diagnostic build (tools/build_variant.sh stamp -DMC_STAMP with
VARIANT_SOURCES="k_mc.hip hmgpu_api.hip"): mean duration of every phase of a wave of k_mc_luma, in shader cycles"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "/tmp/mc_stamps.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] != 0]
a = a[a[:, 7] != 0]
print("waves with stamps:", len(a))
names = ["start -> prologue loads back", "shuffles, addresses, window loads issued", "window loads back", "H compute + LDS writes", "barrier", "V + stores issued", "stores drained"]
d = np.diff(a, axis=1)
for i, n in enumerate(names):
    print("%-44s mean %8.0f  median %8.0f  p90 %8.0f cycles" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
life = a[:, 7] - a[:, 0]
lo, hi = np.percentile(a[:, 0], 0.1), np.percentile(a[:, 7], 99.9)
print("wave lifetime mean %.0f median %.0f; launch span (0.1 .. 99.9 %%) %.0f cycles" % (life.mean(), np.median(life), hi - lo))
if len(sys.argv) > 2:
    ms = float(sys.argv[2])
    print("kernel %.4f ms by events -> s_memtime ticks at %.0f MHz; %.1f waves in flight per CU on average" % (ms, (hi - lo) / ms / 1e3, life.sum() / (hi - lo) / 256))
