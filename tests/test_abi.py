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


def test_pack_levels_keeps_exactly_the_coded_transform_units():
    """hmgpu_pack_levels (host code): the compact stream holds the coded TUs of HM's dense layout, in order, and nothing else"""
    import numpy as np
    import libhm_amd
    from tests import synth
    p = synth.make_picture(416, 240, 10, seed=4, intra_frac=0.2, ref_handles=([0], [0]))
    packed = libhm_amd.pack_levels(p.seq, p.meta, p.coeffs)
    m = p.meta_np
    decoded = p.inside & (m["part_size"] != 8)
    tr = m["tr_idx"]
    log2tu = 6 - m["depth"] - tr
    chain = (1 << (tr + 1)) - 1
    for comp, key in enumerate(("cbf_y", "cbf_u", "cbf_v")):
        coded = decoded & ((m[key] & chain) == chain)
        # coded partitions -> coefficients: 16 per 4x4 luma partition of a coded luma TU; chroma: 4 per partition, except that the four
        # 4x4 luma TUs of an 8x8 node share ONE 4x4 chroma TU (16 coefficients, signalled at the first of them)
        if comp == 0:
            want = int(coded.sum()) * 16
        else:
            big = coded & (log2tu > 2)
            first = coded & (log2tu == 2) & ((np.arange(coded.shape[1])[None, :] & 3) == 0)
            want = int(big.sum()) * 4 + int(first.sum()) * 16
        total = int(packed.starts[comp][-1])
        assert total == want, (comp, total, want)
        dense = p.coeffs.arrays[comp].reshape(-1)
        assert np.abs(packed.arrays[comp][:total].astype(np.int64)).sum() == np.abs(dense.astype(np.int64)).sum()
        assert not packed.arrays[comp][total:].any()
        assert np.all(np.diff(packed.starts[comp].astype(np.int64)) >= 0)
