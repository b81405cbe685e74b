"""Pin the C restatement against picture-level dumps of real HM decodes (fixtures stream_*.npz: per-CTU TComDataCU
metadata + coefficients as HM's parser left them, and HM's planes before deblocking, after deblocking and after SAO)."""
import numpy as np
import pytest

from tests import golden_util as gu


@pytest.mark.parametrize("name", gu.STREAMS + gu.STREAMS_EXT)
def test_reconstruction_matches_hm(oracle, name):
    """every sample of every picture -- inter CUs (MC + residual) and intra CUs (reference samples, smoothing, planar / DC /
    angular prediction, residual, in decoding order) -- recomputed from the parsed data alone"""
    pics = gu.stream_pictures(name)
    finals = []
    n_intra_total = 0
    for p in pics:
        cur = [np.full_like(a, -1) for a in p.pre]
        n_intra = oracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, cur, finals)
        for c in range(3):
            assert np.array_equal(cur[c], p.pre[c]), "%s pic %d comp %d" % (name, p.index, c)
        assert n_intra == int(((p.meta_np["pred_mode"] == 1) & (p.meta_np["part_size"] != 8) & _inside(p)).sum())
        n_intra_total += n_intra
        finals.append(p.fin)
    assert n_intra_total > 0


def test_constrained_intra_fixture_really_is_constrained():
    pics = gu.stream_pictures("ldp_cip_main10_208x120")
    assert all(sl.constrained_intra_pred == 1 for p in pics for sl in p.slices)
    assert pics[0].seq.strong_intra_smoothing == 0
    # P pictures with both intra and inter CUs, otherwise the flag would not matter
    assert any(((p.meta_np["pred_mode"] == 1) & (p.meta_np["part_size"] != 8)).any() and (p.meta_np["pred_mode"] == 0).any()
               for p in pics[1:])


def _inside(p):
    """partitions that lie inside the picture"""
    ok = np.zeros((p.num_ctus, p.parts), dtype=bool)
    for a in range(p.num_ctus):
        cx, cy = (a % p.ctus_w) * p.ctu_size, (a // p.ctus_w) * p.ctu_size
        for z in range(p.parts):
            x = sum(((z >> (2 * b)) & 1) << b for b in range(8))
            y = sum(((z >> (2 * b + 1)) & 1) << b for b in range(8))
            ok[a, z] = (cx + 4 * x < p.width) and (cy + 4 * y < p.height)
    return ok


@pytest.mark.parametrize("name", gu.STREAMS + gu.STREAMS_EXT)
def test_deblocking_matches_hm(oracle, name):
    changed = 0
    for p in gu.stream_pictures(name):
        planes = [a.copy() for a in p.pre]
        oracle.loop_filter_pic(p.seq, p.slices, p.meta, p.pp, planes, 3)
        for c in range(3):
            assert np.array_equal(planes[c], p.dbk[c]), "%s pic %d comp %d" % (name, p.index, c)
        changed += sum(int((p.pre[c] != p.dbk[c]).sum()) for c in range(3))
    if "lossless" not in name:          # (every CU of that stream is exempt from the loop filters: they must change nothing)
        assert changed > 0      # the filter did something somewhere in the stream
    else:
        assert changed == 0


@pytest.mark.parametrize("name", gu.STREAMS + gu.STREAMS_EXT)
def test_sao_matches_hm(oracle, name):
    any_sao = False
    for p in gu.stream_pictures(name):
        rec = oracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
        assert np.array_equal(rec[:, :, :3][rec[:, :, 0] != 0], p.sao_rec[:, :, :3][p.sao_rec[:, :, 0] != 0])
        on = rec[:, :, 0] != 0
        assert np.array_equal(on, p.sao_rec[:, :, 0] != 0)
        assert np.array_equal(rec[on], p.sao_rec[on])
        out = oracle.sao_process(p.seq, p.slices, p.pp, p.meta, rec, p.dbk)
        for c in range(3):
            assert np.array_equal(out[c], p.fin[c]), "%s pic %d comp %d" % (name, p.index, c)
        any_sao |= any(not np.array_equal(p.dbk[c], p.fin[c]) for c in range(3))
        # the reference's own self check (TDecGop.cpp:199-208): MD5 of the final planes
        assert gu.hm_md5(out, [p.bd_y, p.bd_c, p.bd_c]) == p.md5
    assert any_sao or "lossless" in name            # (lossless CUs get their reconstruction back after SAO: nothing may change)


@pytest.mark.parametrize("name", gu.STREAMS + gu.STREAMS_EXT)
def test_picture_hashes_match_hm(name):
    """the oracle's CRC / checksum restatement (TComPicYuvMD5.cpp:87-181) against the values HM computed for its own
    output pictures -- the pin for the device hashes of the GPU suite"""
    from oracle import hmoracle
    for p in gu.stream_pictures(name):
        crc, chk = hmoracle.plane_hashes(p.fin, [p.bd_y, p.bd_c, p.bd_c])
        assert np.array_equal(crc, p.crc), "%s pic %d CRC" % (name, p.index)
        assert np.array_equal(chk, p.checksum), "%s pic %d checksum" % (name, p.index)
