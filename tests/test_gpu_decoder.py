"""End to end on the GPU: an HM-encoded bitstream goes into the libHMDecoder-compatible interface, the host
parses, the device reconstructs, and what comes out of libHMDEC_get_image_plane must be HM's own output pictures, sample for
sample -- with the decoder's own check of the decoded-picture-hash SEI (MD5) green on every picture."""
import numpy as np
import pytest

from libhm_amd import hmdec
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name", gu.STREAMS)
def test_bitstream_to_pictures_matches_hm(name, threads):
    z = gu.load("stream_" + name)
    want = {p.poc: p for p in gu.stream_pictures(name)}
    out = []
    with hmdec.Decoder(threads=threads) as d:
        def on_output(p):
            out.append(p.poc)
            w = want[p.poc]
            for c in range(3):
                assert p.size(c) == (w.fin[c].shape[1], w.fin[c].shape[0])
                assert np.array_equal(p.plane(c), w.fin[c]), "%s POC %d component %d" % (name, p.poc, c)
        d.decode_stream(z["bitstream"], on_output=on_output)
        assert d.pictures_decoded == len(want)
        assert d.hash_mismatches == 0                      # TDecGop.cpp:199-262 against the SEI in the stream
    assert out == sorted(want)                             # every picture, in output order
    assert hmdec.lib().libHMDEC_get_internal_bit_depth(0) == want[out[0]].bd_y


@pytest.mark.parametrize("threads", [1, 3])
def test_change_of_sequence_mid_stream_on_the_device(threads):
    """two HM clips of different geometry back to back (the second IDR activates another SPS: new picture store, new device
    context): every picture of both clips comes out with HM's samples, with parser threads too"""
    names = ("ldp_main8_416x240", "ldp_main10_208x120")
    data = b"".join(bytes(gu.load("stream_" + n)["bitstream"]) for n in names)
    want = [{p.poc: p for p in gu.stream_pictures(n)} for n in names]
    out = []
    with hmdec.Decoder(threads=threads) as d:
        def on_output(p):
            clip = 0 if p.size(0) == (416, 240) else 1
            out.append((clip, p.poc))
            for c in range(3):
                assert np.array_equal(p.plane(c), want[clip][p.poc].fin[c]), "clip %d POC %d component %d" % (clip, p.poc, c)
        d.decode_stream(data, on_output=on_output)
        assert d.hash_mismatches == 0
    assert out == [(0, poc) for poc in sorted(want[0])] + [(1, poc) for poc in sorted(want[1])]


def test_hash_check_notices_a_wrong_picture():
    """the SEI check is live: the same stream with one SEI digest byte flipped reports a mismatch (and still decodes)"""
    z = gu.load("stream_ldp_main8_416x240")
    nals = hmdec.split_nal_units(z["bitstream"])
    bad = []
    hit = False
    for n in nals:
        if not hit and ((n[0] >> 1) & 0x3f) == 40 and n[2] == 132:      # suffix SEI, decoded picture hash
            n = n[:6] + bytes([n[6] ^ 0x55]) + n[7:]
            hit = True
        bad.append(n)
    assert hit
    stream = b"".join(b"\x00\x00\x00\x01" + n for n in bad)
    with hmdec.Decoder() as d:
        d.decode_stream(stream)
        assert d.hash_mismatches == 1
        assert d.pictures_decoded == 3


def test_internals_describe_the_picture():
    """libHMDEC_get_internal_info: blocks tile the picture and carry the parsed values"""
    import ctypes as C
    z = gu.load("stream_ldp_main8_416x240")
    pics = gu.stream_pictures("ldp_main8_416x240")
    L = hmdec.lib()
    with hmdec.Decoder() as d:
        seen = []
        d.decode_stream(z["bitstream"], on_output=lambda p: seen.append(p.poc))
        assert seen == [0, 1, 2]
        p = d.last_decoded()
        assert p.poc == pics[-1].poc
        pm = p.array("pred_mode")
        assert np.array_equal(pm[pics[-1].meta_np["part_size"].reshape(-1) != 8], pics[-1].meta_np["pred_mode"].reshape(-1)[pics[-1].meta_np["part_size"].reshape(-1) != 8])


# 4:2:2 / 4:4:4 variants on the device; the metadata fixtures' streams too (expected: HM's decoder output)
LITE_444 = gu.LITE_EXT + ["stream:" + n for n in gu.STREAMS_EXT]


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("name", gu.LITE + LITE_444 + gu.SURGERY)
def test_syntax_variants_decode_to_the_encoders_reconstruction(name, threads):
    """slices, dependent slice segments, wavefronts, CU-level QP, CRA + leading pictures, 32/16-sample CTUs, conformance window,
    slices of tiles, low-delay B: output == HM's encoder reconstruction (== what HM's decoder must produce), hash SEI check green"""
    if name.startswith("stream:"):
        # a metadata fixture used as a stream: its pictures are HM's decoder output
        pics = {p.poc: p.fin for p in gu.stream_pictures(name[7:])}
        z = {"bitstream": gu.load("stream_" + name[7:])["bitstream"]}
        for poc, fin in pics.items():
            for c in range(3):
                z["poc%02d_%d" % (poc, c)] = fin[c]
        frames = len(pics)
    else:
        z = gu.load("lite_" + name)
        w, h, frames, bd = (int(v) for v in z["geom"])
    out = []
    with hmdec.Decoder(threads=threads) as d:
        def on_output(p):
            out.append(p.poc)
            for c in range(3 if ("poc%02d_1" % p.poc) in z else 1):          # (monochrome: libHM hands out no chroma planes)
                assert np.array_equal(p.cropped_plane(c), z["poc%02d_%d" % (p.poc, c)]), "%s POC %d component %d" % (name, p.poc, c)
        d.decode_stream(z["bitstream"], on_output=on_output)
        assert d.hash_mismatches == 0
        assert d.pictures_decoded == frames
    assert out == list(range(frames))


def test_random_access_at_a_cra_picture():
    """decoding starts at the second CRA picture of the stream: its RASL pictures (which reference pictures before the CRA) are
    dropped (8.1: NoRaslOutputFlag; TDecTop::isRandomAccessSkipPicture), everything else is decoded exactly"""
    z = gu.load("lite_ra_cra_main8_208x120")
    nals = hmdec.split_nal_units(z["bitstream"])
    types = [(n[0] >> 1) & 0x3f for n in nals]
    cras = [i for i, t in enumerate(types) if t == 21]
    assert len(cras) >= 2
    start = cras[1]
    while start > 0 and types[start - 1] in (39, 35):                    # prefix SEI / AUD that belong to the CRA access unit
        start -= 1
    stream = b"".join(b"\x00\x00\x00\x01" + n for n in [n for n, t in zip(nals, types) if t in (32, 33, 34)] + nals[start:])
    out = []
    with hmdec.Decoder() as d:
        def on_output(p):
            out.append(p.poc)
            for c in range(3):
                assert np.array_equal(p.cropped_plane(c), z["poc%02d_%d" % (p.poc, c)]), "POC %d component %d" % (p.poc, c)
        d.decode_stream(stream, on_output=on_output)
        assert d.hash_mismatches == 0
    cra_poc = out[0]
    assert cra_poc in (8, 16) and out == sorted(out) and all(p >= cra_poc for p in out)
    assert len(out) >= 2


@pytest.mark.parametrize("max_layer", [0, 1, 2])
def test_temporal_sub_layers_can_be_dropped(max_layer):
    """libHMDec_set_max_temporal_layer (libHMDecoder.h:133-139): NAL units above the layer are ignored; what remains decodes to
    exactly the same pictures, because a picture never references a higher temporal layer"""
    name = "ra_main10_208x120"
    z = gu.load("stream_" + name)
    want = {p.poc: p for p in gu.stream_pictures(name)}
    layer = {int(z["pic%02d_info" % i][4]): int(z["pic%02d_info" % i][14]) for i in range(int(z["num_pics"][0]))}
    out = []
    with hmdec.Decoder(max_temporal_layer=max_layer) as d:
        def on_output(p):
            out.append(p.poc)
            for c in range(3):
                assert np.array_equal(p.plane(c), want[p.poc].fin[c]), "POC %d component %d" % (p.poc, c)
        d.decode_stream(z["bitstream"], on_output=on_output)
        assert d.hash_mismatches == 0
    assert out == sorted(poc for poc, t in layer.items() if t <= max_layer)
    assert 0 < len(out) < len(want) or max_layer >= max(layer.values())


@pytest.mark.parametrize("stream, pictures, threads", [("bench_ldp_wpp_main10_3840x2160.bin", 5, 8), ("bench_ldp_main10_3840x2160.bin", 5, 1),
                                                       ("bench_ra_main10_1920x1080.bin", 9, 4), ("bench_ldp_main10_1920x1080_17.bin", 17, 3)])
def test_full_size_streams_verify_against_their_hash_sei(stream, pictures, threads):
    """2160p and 1080p HM-encoded clips: every picture's MD5 (computed from the planes that come back from the device) equals the
    decoded-picture-hash SEI HM's encoder put into the stream -- the reference's own end-to-end check (TDecGop.cpp:199-262)"""
    import os
    data = open(os.path.join(gu.GOLD, stream), "rb").read()
    out = []
    with hmdec.Decoder(threads=threads) as d:
        d.decode_stream(data, on_output=lambda p: out.append(p.poc))
        assert d.pictures_decoded == pictures
        assert d.hash_mismatches == 0
    assert out == sorted(out) and len(out) == pictures



@pytest.mark.parametrize("name,threads", [("ra_main10_208x120", 4), ("ra_main10_208x120", 1), ("ldp_main10_208x120", 3)])
def test_decoder_submits_batches_with_compact_levels(name, threads):
    """libhmdec hands the device whole batches through hmgpu_decompress_pictures / hmgpu_filter_pictures (HM: TDecGop.cpp:105,157 once
    per picture): pictures retired together that do not predict from each other -- B pictures of one temporal level of the
    random-access GOP -- share one set of launches; the arrays travel from the pictures' staging blocks, the levels in the compact
    form the parser writes.  The pictures are HM's with the compact form and with the dense one."""
    import os
    z = gu.load("stream_" + name)
    want = {p.poc: p for p in gu.stream_pictures(name)}
    for dense in (False, True):
        os.environ.pop("HMDEC_DENSE_LEVELS", None)
        if dense:
            os.environ["HMDEC_DENSE_LEVELS"] = "1"
        try:
            got = {}
            with hmdec.Decoder(threads=threads) as d:
                d.decode_stream(z["bitstream"], on_output=lambda p: got.update({p.poc: [p.plane(c).copy() for c in range(3)]}))
                assert d.hash_mismatches == 0
                n, batches = d.pictures_decoded, d.device_batches
        finally:
            os.environ.pop("HMDEC_DENSE_LEVELS", None)
        assert n == len(want) and 0 < batches <= n
        for poc, planes in got.items():
            for c in range(3):
                assert np.array_equal(planes[c], want[poc].fin[c]), (dense, poc, c)


@pytest.mark.parametrize("stream, pictures, threads", [("bench_ldp_main10_1920x1080_17.bin", 17, 3), ("bench_ldp_wpp_main10_3840x2160.bin", 5, 8)])
def test_md5_sei_checked_on_the_device(stream, pictures, threads):
    """the decoder's check of the MD5 hash SEI with the chains on the device (hmgpu_picture_hash_begin: no download, no hash threads):
    green on HM's own streams -- more pictures than the ring of hash slots holds -- and red when a picture is wrong"""
    import os
    data = open(os.path.join(gu.GOLD, stream), "rb").read()
    with hmdec.Decoder(threads=threads, device_md5=True) as d:
        d.decode_stream(data)
        assert d.pictures_decoded == pictures
        assert d.hash_mismatches == 0
    # a hash SEI that belongs to another picture: the last byte of every MD5 SEI payload flipped
    nals = hmdec.split_nal_units(data)
    bad, n_sei = [], 0
    for nal in nals:
        b = bytearray(nal)
        if ((b[0] >> 1) & 0x3f) == 40 and len(b) > 20 and b[2] == 132:         # suffix SEI, payload type 132 (decoded picture hash)
            b[6] ^= 0x55
            n_sei += 1
        bad.append(bytes(b))
    assert pictures - 1 <= n_sei <= pictures
    with hmdec.Decoder(threads=threads, device_md5=True) as d:
        d.decode_stream(b"".join(b"\x00\x00\x00\x01" + n for n in bad))
        assert d.hash_mismatches == n_sei


@pytest.mark.parametrize("contexts,threads,device_md5", [(2, 1, True), (2, 4, False), (3, 3, True)])
def test_pictures_placed_on_several_device_contexts(contexts, threads, device_md5, monkeypatch):
    """hmdec_set_devices: the B pictures of one temporal level go round-robin to several device contexts (here all on GPU 0 -- the trick of
    test_picture_transfer_between_contexts; on a node they are one per GPU), every reference picture is copied once to each context
    that predicts from it, and the pictures, their order and the hash SEI checks are what one context gives (TDecTop.cpp:192-213,672;
    TComSlice.cpp:318-376)"""
    import os
    # (pictures that retire alone stay with their first reference; the test makes them move on as well, so that references travel whatever
    # the timing of the parser threads)
    monkeypatch.setenv("HMDEC_PLACE_ROUND_ROBIN", "1")
    name = "ra_main10_208x120"
    z = gu.load("stream_" + name)
    want = {p.poc: p for p in gu.stream_pictures(name)}
    out = []
    with hmdec.Decoder(threads=threads, devices=[0] * contexts, device_md5=device_md5) as d:
        assert d.num_devices == contexts
        def on_output(p):
            out.append(p.poc)
            for c in range(3):
                assert np.array_equal(p.plane(c), want[p.poc].fin[c]), "POC %d component %d" % (p.poc, c)
        d.decode_stream(z["bitstream"], on_output=on_output)
        assert d.pictures_decoded == len(want)
        assert d.hash_mismatches == 0
        moved, batches = d.transfer_bytes, d.device_batches
        assert 0 < batches <= len(want)
    assert out == sorted(want)
    assert moved > 0
    # the full-size random-access clip: hash SEIs green on every picture, same number of pictures, output order kept
    data = open(os.path.join(gu.GOLD, "bench_ra_main10_1920x1080.bin"), "rb").read()
    out = []
    with hmdec.Decoder(threads=threads, devices=[0] * contexts, device_md5=device_md5) as d:
        d.decode_stream(data, on_output=lambda p: out.append(p.poc))
        assert d.pictures_decoded == 9
        assert d.hash_mismatches == 0
        moved = d.transfer_bytes
    with hmdec.Decoder(threads=threads, device_md5=device_md5) as d:      # one context: nothing travels
        d.decode_stream(data)
        assert d.hash_mismatches == 0 and d.transfer_bytes == 0
    assert out == sorted(out) and len(out) == 9 and moved > 0
