"""sha256 over libhm_amd/csrc: ties profiles/hbm_traffic.json to the kernel sources it was measured on (bench.py, tools/pmc_summary.py)"""
import hashlib
import os


def csrc_digest():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libhm_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]
