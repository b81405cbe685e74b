"""libhmdec.so as a drop-in for libHM at source level: a C++ client written against the libHMDecoder interface is compiled against
include/hmdec.h and -- in the builder container, where the reference checkout exists -- against libHM's own libHMDecoder.h, linked
with libhmdec.so and run on a fixture bitstream (parse-only here; the GPU variant checks the first sample against HM's)."""
import os
import subprocess

import numpy as np
import pytest

from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HEADER_DIR = "/root/reference/source/App/libHMDecoder"


def _build(tmp_path, reference_header):
    from libhm_amd import build
    build.build_decoder()
    exe = str(tmp_path / ("client_ref" if reference_header else "client"))
    inc = ["-I" + REF_HEADER_DIR, "-DUSE_REFERENCE_HEADER"] if reference_header else ["-I" + os.path.join(ROOT, "include")]
    libdir = os.path.join(ROOT, "libhm_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "client", "libhm_client.cpp")] + inc +
                          ["-L" + libdir, "-lhmdec", "-lhmgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _run(exe, tmp_path, name, parse_only):
    z = gu.load("stream_" + name)
    bs = tmp_path / (name + ".bin")
    bs.write_bytes(bytes(z["bitstream"]))
    r = subprocess.run([exe, str(bs)] + (["parse-only"] if parse_only else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout.splitlines()


@pytest.mark.parametrize("reference_header", [False, True])
def test_cxx_client_links_and_runs(tmp_path, reference_header):
    if reference_header and not os.path.isdir(REF_HEADER_DIR):
        pytest.skip("the reference checkout is not present on this machine")
    exe = _build(tmp_path, reference_header)
    lines = _run(exe, tmp_path, "ra_main10_208x120", parse_only=True)
    assert lines[0] == "version 16.0"
    pocs = [int(l.split()[1]) for l in lines[1:]]
    assert pocs == list(range(9))
    assert all(" 208x120 chroma 104x60 format 1 depth 10 " in l for l in lines[1:])
    assert all(int(l.split()[-1]) > 0 for l in lines[1:])


def test_corrupted_streams_never_crash(tmp_path):
    """robustness of the host parser: bit flips, overwritten bytes, deletions and insertions anywhere in HM-encoded streams end in a
    decoded stream or in an error code from libHMDec_push_nal_unit (client exit code 4) -- never in a crash or a hang"""
    import random
    exe = _build(tmp_path, False)
    rng = random.Random(20261003)
    names = ["stream_ldp_main8_416x240", "stream_ra_main10_208x120", "lite_ldp_wpp_depslices_main8_416x240", "lite_ldp_tileslices_main10_832x128",
             "stream_ldp_pcm_main8_208x120", "lite_ra_cra_main8_208x120", "stream_ldp_wp_main10_208x120", "stream_ldp_sl_main10_208x120",
             "lite_ldp_wpp_main10_416x240", "lite_ldb_rext420_main8_208x120", "lite_ldb_rext420_lossless_main8_208x120",
             "lite_ldb_rext420_ts32_nosmooth_main8_208x120", "lite_intra_rext420_main8_208x120", "lite_ldb_mono_rext_main8_208x120"]
    outcomes = {0: 0, 4: 0}
    for it in range(120):
        b = bytearray(bytes(gu.load(rng.choice(names))["bitstream"]))
        mode = rng.randrange(4)
        for _ in range(rng.randrange(1, 6)):
            pos = rng.randrange(4, len(b))
            if mode == 0:
                b[pos] ^= 1 << rng.randrange(8)
            elif mode == 1:
                b[pos] = rng.randrange(256)
            elif mode == 2:
                del b[pos:pos + rng.randrange(1, 40)]
            else:
                b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
        f = tmp_path / "fuzz.bin"
        f.write_bytes(bytes(b))
        env = dict(os.environ, HMDEC_THREADS="3") if it % 2 else dict(os.environ)      # every other one with parser threads
        if it % 3 == 0:
            env["HMDEC_CLIENT_KEEP_GOING"] = "1"                                       # ... and some going on after the first error
        r = subprocess.run([exe, str(f), "parse-only"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=60, env=env)
        assert r.returncode in (0, 4), "iteration %d: exit code %d\n%s" % (it, r.returncode, r.stderr[-400:])
        outcomes[r.returncode] += 1
    assert outcomes[0] > 0 and outcomes[4] > 0          # both ends of the spectrum were exercised


@pytest.mark.gpu
def test_cxx_client_on_the_gpu(tmp_path):
    exe = _build(tmp_path, False)
    name = "ldp_main8_416x240"
    lines = _run(exe, tmp_path, name, parse_only=False)
    pics = {p.poc: p for p in gu.stream_pictures(name)}
    assert len(lines) == 1 + len(pics)
    for l in lines[1:]:
        f = l.split()
        assert int(f[f.index("first") + 1]) == int(pics[int(f[1])].fin[0][0, 0])
