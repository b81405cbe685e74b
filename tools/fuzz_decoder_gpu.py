"""robustness run ON the device: corrupted HM-encoded streams through the C++ client (tests/client/libhm_client.cpp) with the GPU
path enabled; every run must end with exit code 0 (decoded) or 4 (error code from libHMDec_push_nal_unit).
usage: python tools/fuzz_decoder_gpu.py <iterations> [seed]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_util as gu
from libhm_amd import build
build.build_decoder()
libdir = os.path.join(ROOT, "libhm_amd")
n, seed = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
names = ["stream_ldp_main8_416x240", "stream_ra_main10_208x120", "lite_ldp_wpp_depslices_main8_416x240", "lite_ldp_tileslices_main10_832x128",
         "stream_ldp_pcm_main8_208x120", "lite_ra_cra_main8_208x120", "stream_ldp_wp_main10_208x120", "stream_ldp_sl_main10_208x120",
         "lite_ldp_slices_main8_208x120", "lite_ldp_dqp_main10_208x120", "lite_ldp_wpp_main10_416x240",
         "lite_ldb_rext420_main8_208x120", "lite_ldb_rext420_lossless_main8_208x120", "lite_ldb_rext420_ts32_nosmooth_main8_208x120",
         "lite_intra_rext420_lossless_main8_208x120", "lite_ldb_rext420_wp_hp_main10_208x120", "lite_ldb_mono_rext_main8_208x120",
         "lite_intra_mono_main8_208x120",
         # round 4: other chroma formats, 12 bits (k_cfmt, k_intra_chroma_422, the unstaged intra paths)
         "stream_ldb_444_ccp_main8_208x120", "stream_ldb_422_main10_208x120", "stream_intra_422_main8_208x120", "lite_ldb_444_main10_208x120",
         "lite_ldb_422_wpp_depslices_main8_416x240", "stream_ldb_main12_208x120", "lite_ldb_444_ccp_main12_208x120", "lite_ldb_ts32_main12_208x120"]
with tempfile.TemporaryDirectory() as tmp:
    exe = os.path.join(tmp, "client")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "client", "libhm_client.cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + libdir, "-lhmdec", "-lhmgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    outcomes = {}
    for it in range(n):
        b = bytearray(bytes(gu.load(rng.choice(names))["bitstream"]))
        mode = rng.randrange(4)
        for _ in range(rng.randrange(1, 6)):
            pos = rng.randrange(4, len(b))
            if mode == 0: b[pos] ^= 1 << rng.randrange(8)
            elif mode == 1: b[pos] = rng.randrange(256)
            elif mode == 2: del b[pos:pos + rng.randrange(1, 40)]
            else: b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
        f = os.path.join(tmp, "f.bin")
        open(f, "wb").write(bytes(b))
        env = dict(os.environ, HMDEC_CLIENT_KEEP_GOING="1")          # an undecodable unit is dropped, the rest of the stream still goes to the device
        if it % 2:
            env["HMDEC_THREADS"] = "3"
        r = subprocess.run([exe, f], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=120, env=env)
        outcomes[r.returncode] = outcomes.get(r.returncode, 0) + 1
        if r.returncode not in (0, 4):
            print("iteration", it, "exit code", r.returncode, r.stderr[-300:])
            sys.exit(1)
    print("fuzz on the device:", n, "runs, outcomes", outcomes)
