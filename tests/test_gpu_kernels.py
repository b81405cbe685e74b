"""GPU parity, kernel level: the HIP inverse-transform and interpolation kernels, called through the C ABI
(hmgpu_inverse_transform_batch / hmgpu_mc_batch), against HM's known answers (tests/golden/kats.npz) and against the
C oracle on seeded random inputs.  Bit-exact (integer pipeline)."""
import numpy as np
import pytest

from libhm_amd import abi
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import libhm_amd
    c = libhm_amd.Context(abi.make_seq(64, 64, 8, max_pictures=1))
    yield c
    c.close()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_inverse_dct_matches_hm_kats(ctx, n, bd):
    z = gu.load("kats")
    coeff = z["itr_in_n%d_bd%d" % (n, bd)]
    log2 = int(np.log2(n))
    # de-quantiser made an identity: scale 64 (rem 4) and right shift 6  =>  per = -(transformShift)
    per = np.full(coeff.shape[0], -(15 - bd - log2), dtype=np.int8)
    rem = np.full(coeff.shape[0], 4, dtype=np.int8)
    got = ctx.inverse_transform_batch(coeff, log2, bd, per, rem, np.zeros(coeff.shape[0], dtype=np.uint8))
    assert np.array_equal(got, z["itr_dct_n%d_bd%d" % (n, bd)])
    if n == 4:
        got = ctx.inverse_transform_batch(coeff, log2, bd, per, rem, np.ones(coeff.shape[0], dtype=np.uint8))
        assert np.array_equal(got, z["itr_dst_n4_bd%d" % bd])


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_dequant_transform_matches_oracle(ctx, oracle, log2, bd):
    rng = np.random.RandomState(100 * bd + log2)
    n = 1 << log2
    ntu = 300
    lev = np.zeros((ntu, n, n), dtype=np.int16)
    # mix: full range stress, sparse typical, all-zero, DC only
    lev[:60] = rng.randint(-32768, 32768, size=(60, n, n))
    k = min(n, 8)
    lev[60:240, :k, :k] = (np.round(rng.laplace(0, 12, size=(180, k, k))) * (rng.rand(180, k, k) < 0.35)).astype(np.int16)
    lev[240:270, 0, 0] = rng.randint(-2000, 2000, size=30)
    qp = rng.randint(0, 52, size=ntu)
    per = np.zeros(ntu, dtype=np.int8)
    rem = np.zeros(ntu, dtype=np.int8)
    for i in range(ntu):
        comp = i % 3
        per[i], rem[i] = oracle.qp_param(int(qp[i]), comp, bd, int(rng.randint(-6, 7)) if comp else 0)
    flags = np.zeros(ntu, dtype=np.uint8)
    if log2 == 2:
        flags[::3] |= 1          # DST
        flags[1::4] |= 2         # transform skip
    want = oracle.inverse_transform_tus(lev, log2, bd, per, rem, flags)
    got = ctx.inverse_transform_batch(lev, log2, bd, per, rem, flags)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("bd", [8, 10])
def test_interpolation_matches_hm_kats(ctx, bd):
    z = gu.load("kats")
    plane = z["interp_plane_bd%d" % bd]
    cases = z["interp_cases_bd%d" % bd]
    out = z["interp_out_bd%d" % bd]
    pos = 0
    for is_chroma in (0, 1):
        for bi in (0, 1):
            sel = [(i, c) for i, c in enumerate(cases) if c[0] == is_chroma and c[1] == bi]
            blocks = np.array([[c[4], c[5], c[6], c[7], c[2], c[3]] for _, c in sel], dtype=np.int32)   # fraction only: mv = frac
            got = ctx.mc_batch(is_chroma, bd, plane, blocks, bi)
            for (i, c), g in zip(sel, got):
                off = int(sum(int(cc[6]) * int(cc[7]) for cc in cases[:i]))
                want = out[off:off + int(c[6]) * int(c[7])].reshape(int(c[7]), int(c[6]))
                assert np.array_equal(g, want), tuple(c)
                pos += 1
    assert pos == len(cases)


@pytest.mark.parametrize("bd", [8, 10])
def test_interpolation_with_border_clamp_matches_oracle(ctx, oracle, bd):
    rng = np.random.RandomState(7 + bd)
    plane = rng.randint(0, 1 << bd, size=(72, 88)).astype(np.int16)
    for is_chroma in (0, 1):
        blocks = []
        for _ in range(120):
            w, h = [(8, 8), (16, 4), (4, 16), (32, 32), (2, 2), (4, 4), (64, 64), (8, 2)][rng.randint(0, 8)]
            x0, y0 = int(rng.randint(0, 88 - 2)), int(rng.randint(0, 72 - 2))
            x0 -= x0 % 2
            y0 -= y0 % 2
            mvx, mvy = int(rng.randint(-80 * 4, 80 * 4)), int(rng.randint(-80 * 4, 80 * 4))
            # what TComDataCU::clipMv guarantees the decoder (CU origin = block origin, luma units; chroma: half of it)
            sc = 2 if is_chroma else 1
            mvx = min(((88 * sc + 8 - x0 * sc - 1) << 2) // 1, max((-64 - 8 - x0 * sc + 1) * 4, mvx))
            mvy = min(((72 * sc + 8 - y0 * sc - 1) << 2) // 1, max((-64 - 8 - y0 * sc + 1) * 4, mvy))
            blocks.append([x0, y0, w, h, mvx, mvy])
        blocks = np.array(blocks, dtype=np.int32)
        for bi in (0, 1):
            got = ctx.mc_batch(is_chroma, bd, plane, blocks, bi)
            for b, g in zip(blocks, got):
                want = oracle.pred_inter_blk(is_chroma, bd, plane, int(b[0]), int(b[1]), int(b[2]), int(b[3]), int(b[4]), int(b[5]), bi)
                assert np.array_equal(g, want), (is_chroma, bi, tuple(b))



@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("is_chroma", [0, 1])
def test_lds_staged_picture_kernels_match_hm_interpolation_kats(bd, is_chroma):
    """HM's own interpolation vectors (kats.npz interp_*: TComInterpolationFilter::filterHor / filterVer called on a 96x96 plane, all
    16 luma and 64 chroma phase pairs, final and 14-bit outputs) through the LDS-staged PICTURE kernels k_mc_luma / k_mc_chroma -- the
    kernels bench.py times -- instead of the test-only flat kernel (tests/kat_pictures.py: a picture of 8x8 CUs, one tile each, PU k
    predicting KAT case k).  P picture: the `uni` cases against HM's final samples; B picture: two `bi` cases per PU, expected =
    addAvg of HM's two 14-bit outputs (the formula itself is checked against HM's addAvg vectors first)."""
    import libhm_amd
    from tests import kat_pictures as kp
    z = gu.load("kats")
    assert np.array_equal(kp.add_avg(z["addavg_a_bd%d" % bd], z["addavg_b_bd%d" % bd], bd), z["addavg_out_bd%d" % bd])
    for bi in (0, 1):
        p, ref, checks = kp.build(bd, is_chroma, bi)
        with libhm_amd.Context(p.seq) as ctx:
            h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
            ctx.upload(h0, ref)
            ctx.upload(h1, ref)
            ctx.upload(hc, [np.zeros_like(a) for a in ref])
            ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
            got = ctx.download(hc)
        kp.verify(got, checks, (bd, is_chroma, bi))
