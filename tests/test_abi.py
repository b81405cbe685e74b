"""The C-ABI library loads and exports every symbol include/hmgpu.h declares; struct mirrors have the C sizes.
(No compute calls: runs without a GPU.)"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    sys.path.insert(0, ROOT)
    from libhm_amd import build
    return build.build()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "hmgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hmgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    names = _declared_functions()
    assert len(names) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    missing = [n for n in names if n not in exported]
    assert not missing, missing


def test_decoder_library_exports_every_declared_symbol(built_lib):
    """libhmdec.so: everything include/hmdec.h declares -- the libHMDecoder-compatible names and this library's own additions"""
    from libhm_amd import build
    dec = build.build_decoder()
    text = open(os.path.join(ROOT, "include", "hmdec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b((?:libHMDec_|libHMDEC_|hmdec_)[A-Za-z0-9_]+)\s*\(", text)))
    assert len(names) >= 28, names
    out = subprocess.check_output(["nm", "-D", "--defined-only", dec], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    # and nothing of the test infrastructure is linked in
    assert "oracle" not in subprocess.check_output(["ldd", dec], text=True)


def test_struct_sizes_match_the_header(built_lib, tmp_path):
    from libhm_amd import abi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "hmgpu.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(hmgpu_seq_params),sizeof(hmgpu_slice_params),sizeof(hmgpu_ctu_meta),sizeof(hmgpu_coeffs),'
                   'sizeof(hmgpu_sao_param),sizeof(hmgpu_pic_params),sizeof(hmgpu_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    mirrors = [abi.SeqParams, abi.SliceParams, abi.CtuMeta, abi.Coeffs, abi.SaoParam, abi.PicParams, abi.Stats]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_product_has_no_oracle_dependency():
    """the product path must not import, link or load anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "libhm_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "hmoracle" not in text and "hm_oracle" not in text and "libhmref" not in text, f
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "libhm_amd", "libhmgpu.so")], text=True)
    assert "oracle" not in out


def test_missing_gpu_fails_loudly(built_lib):
    """without a GPU hmgpu_create must fail with HMGPU_EDEVICE, never fall back to a CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import libhm_amd
    from libhm_amd import abi
    with pytest.raises(libhm_amd.HmgpuError) as e:
        libhm_amd.Context(abi.make_seq(64, 64, 8))
    assert e.value.status == abi.HMGPU_EDEVICE
