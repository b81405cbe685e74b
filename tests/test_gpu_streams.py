"""GPU parity, picture level, through the two drop-in calls of the C ABI (hmgpu_decompress_slice / hmgpu_filter_picture)
on HM's own per-CTU metadata of real decodes (fixtures stream_*.npz), against HM's planes before deblocking, after
deblocking and after SAO, plus the reference's own self check (per-plane MD5 of the final picture)."""
import numpy as np
import pytest

from libhm_amd import abi
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _run_stream(name, check):
    import libhm_amd
    pics = gu.stream_pictures(name)
    with libhm_amd.Context(pics[0].seq) as ctx:
        handles = []
        for p in pics:
            h = ctx.acquire()
            assert h == p.index                 # fixtures map POC -> decode index == device handle
            handles.append(h)
            check(ctx, h, p)
    return len(pics)


# 4:4:4 (with cross-component prediction) and 4:2:2 on the device
STREAMS_444 = gu.STREAMS_EXT


@pytest.mark.parametrize("name", gu.STREAMS + STREAMS_444)
def test_full_chain_matches_hm(name):
    """decompress_slice + filter_picture from the parsed data alone: every sample of every picture, inter and intra CUs
    (I pictures included), is produced on the GPU"""
    def check(ctx, h, p):
        ctx.upload(h, [np.full_like(a, 77) for a in p.pre])          # nothing of HM's picture to start from
        ctx.decompress_slice(h, 0, p.slices[0], p.meta, p.coeffs)
        rec = ctx.download(h)
        for c in range(3):
            assert np.array_equal(rec[c], p.pre[c]), "%s pic %d comp %d: reconstruction" % (name, p.index, c)
        ctx.filter_picture(h, p.pp, p.sao_raw, stages=3)
        dbk = ctx.download(h)
        for c in range(3):
            assert np.array_equal(dbk[c], p.dbk[c]), "%s pic %d comp %d: deblocking" % (name, p.index, c)
        ctx.filter_picture(h, p.pp, p.sao_raw, stages=4)
        fin = ctx.download(h)
        for c in range(3):
            assert np.array_equal(fin[c], p.fin[c]), "%s pic %d comp %d: SAO" % (name, p.index, c)
        assert gu.hm_md5(fin, [p.bd_y, p.bd_c, p.bd_c]) == p.md5   # TDecGop.cpp:199-208
        st = ctx.stats()
        assert st["intra_partitions"] + st["inter_partitions"] > 0
    _run_stream(name, check)


@pytest.mark.parametrize("name", ["ra_main10_208x120", "ldp_main8_416x240"])
def test_single_call_filter_picture_matches_hm(name):
    """the one-call form (stages 7) and the vertical-only / horizontal-only split against the oracle"""
    from oracle import hmoracle
    def check(ctx, h, p):
        ctx.upload(h, p.pre)
        ctx.filter_picture(h, p.pp, p.sao_raw, stages=1)
        got = ctx.download(h)
        want = [a.copy() for a in p.pre]
        hmoracle.loop_filter_pic(p.seq, p.slices, p.meta, p.pp, want, 1)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "vertical edges, pic %d comp %d" % (p.index, c)
        ctx.upload(h, p.pre)
        ctx.filter_picture(h, p.pp, p.sao_raw)
        fin = ctx.download(h)
        for c in range(3):
            assert np.array_equal(fin[c], p.fin[c])
    # metadata must be on the device before filtering: decompress first
    def check2(ctx, h, p):
        ctx.upload(h, p.pre)
        ctx.decompress_slice(h, 0, p.slices[0], p.meta, p.coeffs)
        check(ctx, h, p)
    _run_stream(name, check2)


@pytest.mark.parametrize("name", gu.STREAMS + STREAMS_444)
def test_device_hash_and_packed_output_match_hm(name):
    """f-4: the decoded-picture-hash check (MD5, CRC, checksum) and the output packing (8/16-bit, conformance window) happen on the
    device; the hashes are the ones HM computed, the packed planes are HM's planes cropped (TVideoIOYuv.cpp:706-790)"""
    import libhm_amd
    def check(ctx, h, p):
        ctx.upload(h, [np.full_like(a, 77) for a in p.pre])
        ctx.decompress_slice(h, 0, p.slices[0], p.meta, p.coeffs)
        ctx.filter_picture(h, p.pp, p.sao_raw)
        assert np.array_equal(ctx.picture_hash(h, 2), p.crc), "%s pic %d CRC" % (name, p.index)
        assert np.array_equal(ctx.picture_hash(h, 3), p.checksum), "%s pic %d checksum" % (name, p.index)
        assert bytes(ctx.picture_hash(h, 1)) == p.md5, "%s pic %d MD5" % (name, p.index)     # the chains of k_md5 (TComPicYuvMD5.cpp:183-205)
        nbytes = 1 if p.bd_y <= 8 else 2
        for crop in [(0, 0, 0, 0), (2, 6, 4, 8)]:
            l, r, t, b = crop
            got = ctx.download_packed(h, nbytes, crop)
            for c in range(3):
                sx, sy = (p.csx, p.csy) if c else (0, 0)
                want = p.fin[c][t >> sy:p.fin[c].shape[0] - (b >> sy), l >> sx:p.fin[c].shape[1] - (r >> sx)]
                assert got[c].dtype == (np.uint8 if nbytes == 1 else np.uint16)
                assert np.array_equal(got[c].astype(np.int16), want), "%s pic %d comp %d crop %s" % (name, p.index, c, crop)
        if nbytes == 1:                                              # is16bit output of an 8-bit picture (file bit depth > 8)
            wide = ctx.download_packed(h, 2)
            assert all(np.array_equal(wide[c].astype(np.int16), p.fin[c]) for c in range(3))
        if p.csx:
            with pytest.raises(libhm_amd.HmgpuError):
                ctx.download_packed(h, 1, (1, 0, 0, 0))              # odd window with subsampled chroma
    _run_stream(name, check)


@pytest.mark.parametrize("name", ["ldp_pcm_main8_208x120", "ldp_cip_main10_208x120", "ldp_lossless_main10_208x120", "ldp_tiles_main10_832x128",
                                  "ldp_sl_main10_208x120", "ra_main10_208x120", "ldb_444_ccp_main8_208x120", "ldb_422_main10_208x120", "ldb_main12_208x120"])
def test_inter_pictures_with_intra_cus_in_batches_of_five(name):
    """the P / B pictures of HM's streams -- intra CUs (PCM, lossless, constrained intra prediction, scaling lists, tiles, other chroma formats, 12 bits)
    scattered among inter CUs -- reconstructed FIVE per call: the calls k_intra<1, LEAN> serves (no I slice, mostly inter; one wave per CTU,
    nothing staged in LDS), against HM's reconstruction before the loop filters"""
    import copy
    import libhm_amd
    pics = gu.stream_pictures(name)
    seq = copy.copy(pics[0].seq) if False else type(pics[0].seq).from_buffer_copy(pics[0].seq)
    seq.max_pictures = pics[0].seq.max_pictures + 4
    n_batched = 0
    with libhm_amd.Context(seq) as ctx:
        handles = [ctx.acquire() for _ in pics]
        assert handles == [p.index for p in pics]
        extra = [ctx.acquire() for _ in range(4)]
        for p in pics:
            h = p.index
            junk = [np.full_like(a, 55) for a in p.pre]
            ctx.upload(h, junk)
            if all(int(sl.slice_type) != abi.I_SLICE for sl in p.slices):
                for e in extra:
                    ctx.upload(e, junk)
                ctx.decompress_pictures([(t, p.slices, p.meta, p.coeffs) for t in [h] + extra])
                for t in [h] + extra:
                    rec = ctx.download(t)
                    for c in range(3):
                        assert np.array_equal(rec[c], p.pre[c]), "%s pic %d comp %d (device picture %d)" % (name, p.index, c, t)
                n_batched += 1
            else:
                ctx.decompress_pictures([(h, p.slices, p.meta, p.coeffs)])
            ctx.filter_picture(h, p.pp, p.sao_raw)
            fin = ctx.download(h)
            for c in range(3):
                assert np.array_equal(fin[c], p.fin[c]), "%s pic %d comp %d: final" % (name, p.index, c)
    assert n_batched >= 1
