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
