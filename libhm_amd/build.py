"""Build libhm_amd/libhmgpu.so (HIP kernels + C ABI) for gfx950 with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libhmgpu.so")
SOURCES = ["k_prep.hip", "k_mc.hip", "k_mc_cells.hip", "k_itx.hip", "k_intra.hip", "k_dbk.hip", "k_sao.hip", "k_filter.hip", "k_out.hip", "k_cfmt.hip", "hmgpu_api.hip"]
# Code objects for gfx950 with XNACK (retry on page fault) off, the mode these GPUs run in: with the mode known the compiler schedules loads
# more freely than for "any" (k_mc_luma 0.213 -> 0.208 ms per launch of 16 pictures, measured A/B/A/B on one box).  A device that runs with
# HSA_XNACK=1 does not load them: HMGPU_XNACK_ANY=1 in the environment of the build gives the mode-agnostic objects back.
ARCH = ["--offload-arch=gfx950" if os.environ.get("HMGPU_XNACK_ANY") == "1" else "--offload-arch=gfx950:xnack-"]
FLAGS = ARCH + ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "hmgpu_dev.h"), os.path.join(CSRC, "itx_core.h"), os.path.join(CSRC, "mc_core.h"), os.path.join(CSRC, "filter_core.h"), os.path.join(os.path.dirname(HERE), "include", "hmgpu.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("libhmgpu.so build failed")
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc] + ARCH + ["-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


DEC = os.path.join(HERE, "dec")
DEC_OUT = os.path.join(HERE, "libhmdec.so")
DEC_SOURCES = ["params.cpp", "cabac.cpp", "slice_decoder.cpp", "decoder.cpp", "facade.cpp"]


def build_decoder(force=False, verbose=False):
    """libhm_amd/libhmdec.so: the host parser + libHMDecoder-compatible interface (plain C++, links libhmgpu.so)."""
    gpu = build(force=False, verbose=verbose)
    cxx = os.environ.get("CXX", "g++")
    srcs = [os.path.join(DEC, f) for f in DEC_SOURCES]
    deps = srcs + [os.path.join(DEC, f) for f in os.listdir(DEC) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(HERE), "include", "hmgpu.h"), os.path.join(os.path.dirname(HERE), "include", "hmdec.h"), gpu]
    if force or _stale(DEC_OUT, deps):
        cmd = [cxx, "-std=c++17", "-O3", "-fPIC", "-shared", "-pthread", "-Wall", "-o", DEC_OUT] + srcs + [
            "-L" + HERE, "-lhmgpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return DEC_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_decoder(force="--force" in sys.argv, verbose=True))
