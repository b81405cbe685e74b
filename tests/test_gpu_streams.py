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


@pytest.mark.parametrize("name", gu.STREAMS)
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
