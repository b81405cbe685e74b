"""Host parser (libhmdec.so) against HM: every fixture stream is an HM-encoded bitstream together with what HM's
own parser left behind for each picture (per-CTU TComDataCU arrays, coefficient levels, SAO parameters, slice constants, dumped by
oracle/ref_harness.cpp).  The parser runs without a device (parse-only) and must reproduce all of it, array for array."""
import ctypes as C

import numpy as np
import pytest

from libhm_amd import abi, hmdec
from tests import golden_util as gu

ARRAYS = [("depth", "depth"), ("part_size", "part_size"), ("pred_mode", "pred_mode"), ("qp", "qp"), ("tr_idx", "tr_idx"),
          ("cbf0", "cbf_y"), ("cbf1", "cbf_u"), ("cbf2", "cbf_v"), ("ts0", "ts_y"), ("ts1", "ts_u"), ("ts2", "ts_v"),
          ("intra_dir0", "intra_dir_l"), ("intra_dir1", "intra_dir_c"), ("ref_idx0", "ref_idx0"), ("ref_idx1", "ref_idx1"),
          ("mv0", "mv0"), ("mv1", "mv1"), ("bypass", "bypass"), ("ipcm", "ipcm")]


def _decode(name, **kw):
    """everything the parser produced, per picture in HM's decoding order (the fixture's order).  The arrays are taken when a picture
    is put out: that works the same with parser threads, where pictures leave the pipeline in bursts."""
    z = gu.load("stream_" + name)
    by_poc, outputs = {}, []
    with hmdec.Decoder(parse_only=True, **kw) as d:
        def on_output(p):
            arrays = {n: p.array(n) for n, _ in ARRAYS}
            for n in ("skip", "merge", "coeff0", "coeff1", "coeff2", "sao", "slice_idx", "tile_idx", "pcm0", "pcm1", "pcm2"):
                arrays[n] = p.array(n)
            if p.geometry()["chroma_format"] == 3:
                arrays["ccp0"], arrays["ccp1"] = p.array("ccp0"), p.array("ccp1")
            by_poc[p.poc] = dict(poc=p.poc, arrays=arrays, slices=[p.slice_params(i) for i in range(p.num_slices())], hash=p.hash_sei())
            outputs.append(p.poc)
        d.decode_stream(z["bitstream"], on_output=on_output)
    order = [int(z["pic%02d_info" % i][4]) for i in range(int(z["num_pics"][0]))]
    assert sorted(order) == sorted(by_poc), (order, sorted(by_poc))
    return z, [by_poc[poc] for poc in order], outputs


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", gu.STREAMS + gu.STREAMS_EXT)
def test_parser_reproduces_hm_metadata(name, threads):
    z, got, outputs = _decode(name, threads=threads)
    pics = gu.stream_pictures(name)
    assert len(got) == len(pics)
    for g, p in zip(got, pics):
        assert g["poc"] == p.poc                                                 # decoding order and POC (8.3.1)
        decoded = p.meta_np["part_size"].reshape(-1) != abi.SIZE_NONE           # partitions inside the picture
        k = "pic%02d_" % p.index
        for mine, theirs in ARRAYS:
            a, b = g["arrays"][mine], p.meta_np[theirs].reshape(-1)
            if mine.startswith("mv"):
                a, b = a.reshape(-1, 2), p.meta_np[theirs].reshape(-1, 2)
            assert np.array_equal(a[decoded], b[decoded]), "%s pic %d: %s" % (name, p.index, mine)
        assert np.array_equal(g["arrays"]["skip"][decoded], z[k + "meta_skip"].reshape(-1)[decoded])
        inter = decoded & (p.meta_np["pred_mode"].reshape(-1) == abi.MODE_INTER)
        assert np.array_equal(g["arrays"]["merge"][inter], z[k + "meta_merge"].reshape(-1)[inter])
        for c in range(3):
            assert np.array_equal(g["arrays"]["coeff%d" % c], z[k + "coeff%d" % c].reshape(-1)), "%s pic %d: levels %d" % (name, p.index, c)
        assert np.array_equal(g["arrays"]["slice_idx"], p.meta_np["slice_idx"])
        if "tile_idx" in p.meta_np:
            assert np.array_equal(g["arrays"]["tile_idx"], p.meta_np["tile_idx"])
        if "ccp_u" in p.meta_np:                                                 # cross-component prediction weights (4:4:4)
            for c, key in enumerate(("ccp_u", "ccp_v")):
                assert np.array_equal(g["arrays"]["ccp%d" % c][decoded], p.meta_np[key].reshape(-1)[decoded]), "%s pic %d: %s" % (name, p.index, key)
        if (k + "pcm0") in z.files and p.meta_np["ipcm"].any():
            ipcm = p.meta_np["ipcm"].reshape(p.num_ctus, -1)
            for c in range(3):
                per = 16 >> (2 if c else 0)
                mask = np.repeat(ipcm, per, axis=1).reshape(-1).astype(bool)          # HM layout: 16 (4) samples per partition
                assert np.array_equal(g["arrays"]["pcm%d" % c][mask], z[k + "pcm%d" % c].reshape(-1)[mask])


@pytest.mark.parametrize("name", gu.STREAMS + gu.STREAMS_EXT)
def test_parser_reproduces_hm_sao_and_slice_constants(name):
    z, got, outputs = _decode(name)
    pics = gu.stream_pictures(name)
    for g, p in zip(got, pics):
        sao = g["arrays"]["sao"].reshape(p.num_ctus, 3, 35)
        want = p.sao_raw
        assert np.array_equal(sao[:, :, 0], want[:, :, 0]), "SAO mode, pic %d" % p.index
        on = want[:, :, 0] != 0
        assert np.array_equal(sao[:, :, 1][on], want[:, :, 1][on])                          # EO class / BO / merge direction
        new = want[:, :, 0] == 1
        bo = new & (want[:, :, 1] == 4)
        assert np.array_equal(sao[:, :, 2][bo], want[:, :, 2][bo])                          # band position
        assert np.array_equal(sao[bo][:, 3:], want[bo][:, 3:])
        eo = new & (want[:, :, 1] < 4)
        assert np.array_equal(sao[eo][:, 3:8], want[eo][:, 3:8])
        assert len(g["slices"]) == len(p.slices)
        for (mine, lists), theirs in zip(g["slices"], p.slices):
            for f in ("slice_type", "cb_qp_offset", "cr_qp_offset", "pps_cb_qp_offset", "pps_cr_qp_offset", "deblocking_disable",
                      "lf_across_slices", "weighted_pred", "lf_across_tiles", "constrained_intra_pred"):
                assert getattr(mine, f) == getattr(theirs, f), "%s pic %d: %s" % (name, p.index, f)
            if not theirs.deblocking_disable:
                assert (mine.beta_offset_div2, mine.tc_offset_div2) == (theirs.beta_offset_div2, theirs.tc_offset_div2)
            for l in range(2):
                assert mine.num_ref_idx[l] == theirs.num_ref_idx[l]
                for i in range(mine.num_ref_idx[l]):
                    assert mine.ref_poc[l][i] == theirs.ref_poc[l][i]
            if theirs.weighted_pred:
                assert list(mine.wp_log2_denom) == list(theirs.wp_log2_denom)
                for l in range(2):
                    for i in range(mine.num_ref_idx[l]):
                        assert list(mine.wp_weight[l][i]) == list(theirs.wp_weight[l][i]), (l, i)
                        assert list(mine.wp_offset[l][i]) == list(theirs.wp_offset[l][i]), (l, i)
            assert bool(mine.scaling_lists) == (p.scaling_lists is not None)
            if p.scaling_lists is not None:
                for sz in range(4):
                    for l in range(6):
                        if sz == 3 and l % 3:
                            continue                                                          # 32x32 chroma: not used in 4:2:0
                        n = 16 if sz == 0 else 64
                        assert list(lists.coef[sz][l][:n]) == list(p.scaling_lists.coef[sz][l][:n]), (sz, l)
                        if sz >= 2:
                            assert lists.dc[sz][l] == p.scaling_lists.dc[sz][l]


@pytest.mark.parametrize("name", ["ra_main10_208x120", "ldp_main8_416x240", "ra_wp_main8_208x120"])
def test_output_order_and_hash_sei(name):
    """pictures leave in increasing POC order (libHMDecoder.h:183-187), all of them, and the decoded-picture-hash SEI of every
    picture is the MD5 HM computed for its own reconstruction (the fixtures were encoded with SEIDecodedPictureHash=1)"""
    z, got, outputs = _decode(name)
    pics = gu.stream_pictures(name)
    assert outputs == sorted(p.poc for p in pics)
    for g, p in zip(got, pics):
        method, digest = g["hash"]
        assert method == 1 and digest == p.md5


def test_split_nal_units_and_version():
    assert hmdec.lib().libHMDec_get_version() == b"16.0"
    units = hmdec.split_nal_units(b"\x00\x00\x00\x01\x40\x01\xaa\x00\x00\x01\x42\x01\xbb\xcc\x00\x00\x00\x01\x44\x01")
    assert units == [b"\x40\x01\xaa", b"\x42\x01\xbb\xcc", b"\x44\x01"]


def test_errors_are_reported_not_fatal():
    with hmdec.Decoder(parse_only=True) as d:
        with pytest.raises(RuntimeError):
            d.push(b"\x80\x01\x00\x00", False)             # forbidden_zero_bit
        with pytest.raises(RuntimeError):
            d.push(b"\x02\x01\x80\x00\x00", False)          # a slice before any parameter set


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("name", gu.LITE + gu.SURGERY)
def test_parser_gets_through_every_syntax_variant(name, threads):
    """slices, dependent slice segments, wavefronts, CU-level QP, CRA with leading pictures, 32/16-sample CTUs, cropping, slices of
    tiles, B low delay: the parser stays in sync to the last bit of every slice (the stop bit check behind end_of_slice_segment_flag),
    puts all pictures out in POC order, and the hash SEIs it collected are the MD5s of HM's encoder reconstruction"""
    import hashlib
    z = gu.load("lite_" + name)
    w, h, frames, bd = (int(v) for v in z["geom"])
    hashes, out = {}, []
    with hmdec.Decoder(parse_only=True, threads=threads) as d:
        depths = {}
        d.decode_stream(z["bitstream"], on_output=lambda p: (hashes.__setitem__(p.poc, p.hash_sei()), out.append(p.poc),
                                                             depths.update(y=p.geometry()["bd_y"], c=p.geometry()["bd_c"])))
        assert d.pictures_decoded == frames
    assert out == list(range(frames))
    if "crop" in name or "surgery" in name:
        return                                                   # the SEI hashes the uncropped picture, the fixture holds the cropped one; rewritten streams carry no SEI
    for poc in range(frames):
        method, digest = hashes[poc]
        ncomp = 3 if ("poc%02d_1" % poc) in z else 1                         # (monochrome: one digest)
        want = b"".join(hashlib.md5(z["poc%02d_%d" % (poc, c)].astype(np.uint8 if depths["c" if c else "y"] <= 8 else "<u2").tobytes()).digest()
                        for c in range(ncomp))
        assert method == 1 and digest[:16 * ncomp] == want, "POC %d" % poc


def test_internal_info_tiles_the_picture():
    """libHMDEC_get_internal_info (libHMDecoder.cpp:602-720): CU blocks tile the picture exactly, PU / TU entries carry the parsed
    values of the block they name"""
    name = "ra_main10_208x120"
    z = gu.load("stream_" + name)
    pics = gu.stream_pictures(name)
    with hmdec.Decoder(parse_only=True) as d:
        done = []
        def on_decoded(p):
            ref = pics[len(done)]
            done.append(p.poc)
            cover = np.zeros((ref.height, ref.width), dtype=np.int32)
            for x, y, w, h, v, _ in d.internal_info(p, "CU_PREDICTION_MODE"):
                cover[y:y + h, x:x + w] += 1
                assert v in (0, 1)
            assert np.all(cover == 1)
            depth = p.array("depth")
            skips = d.internal_info(p, "CU_SKIP_FLAG")
            assert len(skips) == len(d.internal_info(p, "CU_PART_MODE"))
            mv0 = p.array("mv0").reshape(-1, 2)
            n_mv = 0
            for x, y, w, h, v, v2 in d.internal_info(p, "PU_MV_0"):
                ctu = (y // 64) * ref.ctus_w + x // 64
                inside = [(yy, xx) for yy in range(y, y + h, 4) for xx in range(x, x + w, 4)]
                # every 4x4 partition of the PU holds the PU's vector
                for yy, xx in inside[:1] + inside[-1:]:
                    bx, by = (xx % 64) // 4, (yy % 64) // 4
                    part = ctu * 256 + sum((((bx >> b) & 1) << (2 * b)) | (((by >> b) & 1) << (2 * b + 1)) for b in range(4))
                    assert (mv0[part][0], mv0[part][1]) == (v, v2)
                n_mv += 1
            assert (n_mv > 0) == (ref.slice_type != 2)
            cbf = d.internal_info(p, "TU_CBF_Y")
            assert all(v in (0, 1) for *_, v, _ in cbf) and len(cbf) > 0
            assert [b[4] for b in d.internal_info(p, "CTU_SLICE_INDEX")] == [0] * ref.num_ctus
        d.decode_stream(z["bitstream"], on_decoded=on_decoded)
        assert len(done) == len(pics)


@pytest.mark.parametrize("stream, threads", [("bench_ldp_wpp_main10_3840x2160.bin", 8), ("bench_ra_main10_1920x1080.bin", 5)])
def test_parser_threads_change_nothing_at_full_size(stream, threads):
    """frame-parallel and row-parallel parsing against the single-threaded parser on full-size HM-encoded clips (2160p with wavefronts:
    34 CTB rows per picture shared among the threads; 1080p random access: pictures overlapping): every array of every picture equal"""
    import os
    data = open(os.path.join(gu.GOLD, stream), "rb").read()
    names = [n for n, _ in ARRAYS] + ["skip", "merge", "coeff0", "coeff1", "coeff2", "sao", "slice_idx"]

    def run(t):
        out = {}
        with hmdec.Decoder(parse_only=True, threads=t) as d:
            def on_output(p):
                import zlib
                out[p.poc] = {n: zlib.crc32(p.array(n).tobytes()) for n in names}
                out[p.poc]["hash"] = p.hash_sei()
            d.decode_stream(data, on_output=on_output)
        return out
    a, b = run(1), run(threads)
    assert sorted(a) == sorted(b) and len(a) >= 5
    for poc in a:
        assert a[poc] == b[poc], "POC %d: %s" % (poc, [n for n in a[poc] if a[poc][n] != b[poc][n]])


@pytest.mark.parametrize("threads", [1, 3])
def test_change_of_sequence_mid_stream(threads):
    """two clips of different geometry back to back: the second IDR activates another SPS, the picture store starts over -- after
    every picture of the first clip has come out, with parser threads too (the unit that opens the new sequence is not taken in
    before the application has flushed: it used to free pictures that were still queued for output)"""
    a, b = gu.load("stream_ldp_main8_416x240"), gu.load("stream_ldp_main10_208x120")
    data = bytes(a["bitstream"]) + bytes(b["bitstream"])
    na, nb = int(a["num_pics"][0]), int(b["num_pics"][0])
    out = []
    with hmdec.Decoder(parse_only=True, threads=threads) as d:
        d.decode_stream(data, on_output=lambda p: out.append((p.poc, p.geometry()["width"], p.geometry()["height"], p.hash_sei())))
        assert d.pictures_decoded == na + nb
    assert [(w, h) for _, w, h, _ in out] == [(416, 240)] * na + [(208, 120)] * nb
    assert [poc for poc, _, _, _ in out[:na]] == sorted(poc for poc, _, _, _ in out[:na])
    assert [poc for poc, _, _, _ in out[na:]] == sorted(poc for poc, _, _, _ in out[na:])
    ref = {}
    for z, n in ((a, na), (b, nb)):
        with hmdec.Decoder(parse_only=True) as d:
            d.decode_stream(z["bitstream"], on_output=lambda p: ref.setdefault((p.geometry()["width"], p.poc), p.hash_sei()))
    for poc, w, h, hs in out:
        assert hs == ref[(w, poc)]
