"""HM's interpolation known answers (tests/golden/kats.npz interp_*) laid out as PICTURES: a picture of 8x8 CUs whose PU k carries the
motion vector that makes its prediction the top-left block of KAT case k, so that the vectors reach the picture-level paths (the
oracle's decompress_ctus on the CPU, the LDS-staged k_mc_luma / k_mc_chroma on the GPU) and not only the block-level seams."""
import numpy as np

from libhm_amd import abi
from tests import golden_util as gu
from tests import synth


def add_avg(a, b, bd):
    """TComYuv::addAvg (TComYuv.cpp:336-391) on HM's 14-bit intermediates"""
    shift = 15 - bd
    return np.clip((a.astype(np.int32) + b.astype(np.int32) + (1 << (shift - 1)) + 2 * 8192) >> shift, 0, (1 << bd) - 1)


def build(bd, is_chroma, bi):
    """-> (SynthPicture, reference planes, checks) with checks = [(comp, y, x, expected block, case a, case b or None)]"""
    z = gu.load("kats")
    plane, cases, out = z["interp_plane_bd%d" % bd], z["interp_cases_bd%d" % bd], z["interp_out_bd%d" % bd]
    offs = np.concatenate([[0], np.cumsum(cases[:, 6].astype(np.int64) * cases[:, 7])])
    cs = 1 if is_chroma else 0
    size = 128 << cs            # luma picture size (whole CTUs): the KAT plane is the top-left 96x96 of the luma plane, or of both chroma planes
    blk = 8 >> cs               # samples of the component under one 8x8 PU
    sel = np.array([i for i, c in enumerate(cases) if c[0] == is_chroma and c[1] == bi])
    p = synth.make_picture(size, size, bd, seed=3, bi=bool(bi), mode_probs=(0, 0, 0, 1.0, 0), cbf_prob=0.0, sao=False, ref_handles=([0], [1]))
    m = dict(p.meta_np)
    cu = (p.py >> 3) * (size >> 3) + (p.px >> 3)           # index of the 8x8 CU of every partition
    pux, puy = (p.px & ~7) >> cs, (p.py & ~7) >> cs        # origin of the partition's PU in the component (one vector per PU: TComDataCU)
    for name, ks in (("mv0", sel[cu % len(sel)]), ("mv1", sel[(cu * 7 + 3) % len(sel)])):
        # integer part: from the PU to the KAT block; fraction in the component's own units = quarter luma samples either way
        mvx = ((cases[ks, 4] - pux) << (2 + cs)) + cases[ks, 2]
        mvy = ((cases[ks, 5] - puy) << (2 + cs)) + cases[ks, 3]
        m[name] = np.stack([mvx, mvy], axis=2).astype(np.int16)
    m["ref_idx0"] = np.where(p.inside, 0, -1)
    m["ref_idx1"] = np.where(p.inside, 0 if bi else -1, -1)
    if not bi:
        m["mv1"] = np.zeros_like(m["mv1"])
    p.meta = abi.MetaHolder(m)
    ref = [np.full((size >> (1 if c else 0),) * 2, 17 * c + 5, dtype=np.int16) for c in range(3)]
    for c in ((1, 2) if is_chroma else (0,)):
        ref[c][:96, :96] = plane

    def kat(k):
        c = cases[k]
        return out[offs[k]:offs[k + 1]].reshape(int(c[7]), int(c[6]))
    checks = []
    for y8 in range(size >> 3):
        for x8 in range(size >> 3):
            n = y8 * (size >> 3) + x8
            ka, kb = int(sel[n % len(sel)]), int(sel[(n * 7 + 3) % len(sel)])
            a = kat(ka)
            hh, ww = min(blk, a.shape[0]), min(blk, a.shape[1])
            if bi:
                b = kat(kb)
                hh, ww = min(hh, b.shape[0]), min(ww, b.shape[1])
                want = add_avg(a[:hh, :ww], b[:hh, :ww], bd)
            else:
                want = a[:hh, :ww]
            for comp in ((1, 2) if is_chroma else (0,)):
                checks.append((comp, y8 * blk, x8 * blk, want, ka, kb if bi else None))
    assert {c[4] for c in checks} == set(int(v) for v in sel)
    return p, ref, checks


def verify(got, checks, tag):
    z = gu.load("kats")
    for comp, y, x, want, ka, kb in checks:
        g = got[comp][y:y + want.shape[0], x:x + want.shape[1]]
        assert np.array_equal(g, want), (tag, comp, y, x, ka, kb)
