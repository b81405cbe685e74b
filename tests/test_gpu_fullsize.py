"""GPU parity at BASELINE.json's full sizes on synthetic HM-shaped pictures (tests/synth.py): bit-exact against the C
oracle (which finishes a 2160p picture in well under a second), plus size-independent properties of the path."""
import numpy as np
import pytest

from libhm_amd import abi
from tests import golden_util as gu
from tests import synth

pytestmark = pytest.mark.gpu


def _oracle_chain(oracle, p, cur, refs):
    rec = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, [p.slice], p.meta, p.coeffs, rec, refs)
    dbk = [a.copy() for a in rec]
    oracle.loop_filter_pic(p.seq, [p.slice], p.meta, p.pp, dbk, 3)
    prm = oracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
    fin = oracle.sao_process(p.seq, [p.slice], p.pp, p.meta, prm, dbk)
    return rec, dbk, fin


@pytest.mark.parametrize("bi", [False, True])
def test_explicit_weighted_prediction_matches_oracle(oracle, bi):
    """explicit weighted prediction (TComWeightPrediction.cpp: weightUnidir / weightBidir, per reference index and component) on a picture
    large enough for every PU shape and every residual mask: the k_mc_*<WP, BI> variants, which also add the residual, against the
    oracle.  (The HM-made weighted-prediction streams are 208x120.)"""
    import libhm_amd
    width, height, bd = 1280, 704, 10
    p = synth.make_picture(width, height, bd, seed=77 + int(bi), bi=bi, intra_frac=0.1, ref_handles=([0], [1]))
    sl = p.slice
    sl.weighted_pred = 1
    sl.wp_log2_denom[0], sl.wp_log2_denom[1] = 5, 4
    rng = np.random.RandomState(5)
    for l in range(2):
        for r in range(1):
            for c in range(3):
                sl.wp_weight[l][r][c] = int((1 << sl.wp_log2_denom[1 if c else 0]) + rng.randint(-12, 13))
                sl.wp_offset[l][r][c] = int(rng.randint(-20, 21))
    ref0 = synth.noise_planes(width, height, bd, 21)
    ref1 = synth.blocky_planes(width, height, bd, 22)
    cur = synth.blocky_planes(width, height, bd, 23)
    want_rec, want_dbk, want_fin = _oracle_chain(oracle, p, cur, [ref0, ref1])
    with libhm_amd.Context(p.seq) as ctx:
        h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(h1, ref1)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, sl, p.meta, p.coeffs)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_rec[c]), "reconstruction comp %d" % c
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_fin[c]), "filtered comp %d" % c


@pytest.mark.parametrize("width,height,bd,bi,intra", [(3840, 2160, 10, False, 0.1), (3840, 2160, 10, True, 0.0),
                                                      (3840, 2160, 10, True, 0.25),      # the slowest filter configuration: bi-pred + Bs 2 edges
                                                      (1920, 1080, 10, False, 0.05), (416, 240, 8, True, 0.1),
                                                      (1920, 1080, 10, False, 1.0), (832, 480, 8, False, 0.5),
                                                      (1920, 1080, 12, True, 0.1), (832, 480, 12, False, 0.3)])      # 12 bits: head room 2, 32-bit chroma H pass
def test_synthetic_picture_matches_oracle(oracle, width, height, bd, bi, intra):
    import libhm_amd
    p = synth.make_picture(width, height, bd, seed=width + bd + int(bi), bi=bi, intra_frac=intra, ref_handles=([0], [1]))
    ref0 = synth.noise_planes(width, height, bd, 11)
    ref1 = synth.blocky_planes(width, height, bd, 12)
    cur = synth.blocky_planes(width, height, bd, 13)          # overwritten everywhere: inter AND intra CUs are reconstructed
    want_rec, want_dbk, want_fin = _oracle_chain(oracle, p, cur, [ref0, ref1])
    with libhm_amd.Context(p.seq) as ctx:
        h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(h1, ref1)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_rec[c]), "reconstruction comp %d" % c
        ctx.filter_picture(hc, p.pp, p.sao_raw, stages=3)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_dbk[c]), "deblocking comp %d" % c
        ctx.filter_picture(hc, p.pp, p.sao_raw, stages=4)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_fin[c]), "SAO comp %d" % c
        st = ctx.stats()
        assert st["intra_partitions"] == int(p.intra.sum())
        assert st["inter_partitions"] == int((p.inside & ~p.intra).sum())
        # the same picture again with all three loop-filter stages in ONE call: the fused kernel (what bench.py times)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_fin[c]), "fused loop filter comp %d" % c


@pytest.mark.parametrize("width,height,bd,mode_probs,tr_split,intra,cip,slices,seed", [
    (832, 480, 8, (0, 0, 0, 1, 0), 1.0, 1.0, 0, 1, 1),          # every 8x8 CU split into 4x4 TUs: groups of four 4x4 luma TUs everywhere
    (832, 480, 10, (0, 0, 0.5, 0.5, 0), 0.6, 1.0, 0, 1, 2),     # 16x16 / 8x8 CUs, 8x8 and 4x4 TUs mixed
    (200, 136, 8, (0, 0, 0, 1, 0), 0.7, 1.0, 0, 1, 3),          # partial CTUs on both borders (picture sizes are multiples of 8 only)
    (832, 480, 10, (0.1, 0.3, 0.3, 0.2, 0.1), 0.5, 0.5, 1, 1, 4),   # constrained intra prediction beside inter CUs: availability with holes
    (832, 480, 8, (0, 0.2, 0.4, 0.4, 0), 0.5, 1.0, 0, 5, 5),    # five slices starting mid-row: no references across their borders
    (1920, 1080, 10, (0, 0, 0.3, 0.7, 0), 0.8, 1.0, 1, 3, 6)])
def test_intra_scheduler_paths(oracle, width, height, bd, mode_probs, tr_split, intra, cip, slices, seed):
    """the shapes k_intra's scheduler treats specially -- groups of four 4x4 luma TUs run as one entry, availability masks worked out
    per TU when the CTU starts, reference padding as an index clamp or, where the available units have holes (constrained intra
    prediction, slice borders), bit scans, residual from k_itx's tiles -- against the oracle; single call, then batch entry with compact levels"""
    import libhm_amd
    p = synth.make_picture(width, height, bd, seed=0x1A7 + seed, mode_probs=mode_probs, tr_split_prob=tr_split, intra_frac=intra, cbf_prob=0.7,
                           sao=False, ref_handles=([0], [0]), num_slices=slices)
    for sl in p.slices:
        sl.constrained_intra_pred = cip
    ref0 = synth.noise_planes(width, height, bd, 31)
    cur = synth.blocky_planes(width, height, bd, 33)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, want, [ref0])
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc = ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(hc, cur)
        ctx.decompress_pictures([(hc, p.slices, p.meta, p.coeffs)])
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "reconstruction comp %d" % c
        ctx.upload(hc, cur)
        ctx.decompress_pictures([(hc, p.slices, p.meta, ctx.pack_levels(p.meta, p.coeffs))])
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "compact levels, comp %d" % c


def test_intra_wavefront_gives_up_instead_of_hanging(oracle):
    """the bounded wait of k_intra (a CTU whose neighbour never finishes flags the picture instead of spinning for ever): with the test
    hook that leaves one CTU of an all-intra picture out, the next synchronisation returns HMGPU_EDEVICE with device error -2 (never a
    hang, never a silently wrong picture); switched off again, the same picture reconstructs bit-exactly"""
    import libhm_amd
    width, height, bd = 416, 240, 10
    p = synth.make_picture(width, height, bd, seed=0xFA17, intra_frac=1.0, cbf_prob=0.5, sao=False, ref_handles=([0], [0]))
    cur = synth.blocky_planes(width, height, bd, 5)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, want, [cur])
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc = ctx.acquire(), ctx.acquire()
        ctx.upload(h0, cur)
        ctx.upload(hc, cur)
        ctx.debug_stall_intra(hc, 1)                               # CTU 1: its right and lower neighbours wait for it
        ctx.decompress_slice(hc, 0, p.slices[0], p.meta, p.coeffs)
        with pytest.raises(libhm_amd.HmgpuError) as e:
            ctx.sync()
        assert e.value.status == 2 and e.value.device_error == -2      # HMGPU_EDEVICE, "an intra wavefront gave up waiting"
        ctx.debug_stall_intra(hc, -1)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slices[0], p.meta, p.coeffs)
        got = ctx.download(hc)                                     # (and the flag was cleared with the report)
        for c in range(3):
            assert np.array_equal(got[c], want[c])


@pytest.mark.parametrize("dist", ["stress", "dense"])
@pytest.mark.parametrize("mode_probs,tr_split,intra,ts", [
    ((1.0, 0, 0, 0, 0), 0.35, 0.0, False),      # bench.py --workload idct: 64x64 CUs = four 32x32 luma + 16x16 chroma TUs each
    ((0, 0, 0, 1.0, 0), 1.0, 0.0, True),        # 8x8 CUs split once: 4x4 luma TUs, the shared 4x4 chroma TU, transform skip on half of them
    ((0, 0, 0.5, 0.5, 0), 0.5, 0.3, True)])     # 16x16 / 8x8 / 4x4 TUs; intra CUs (DST, k_intra) beside inter CUs
def test_full_range_levels_through_production_residual_path(oracle, mode_probs, tr_split, intra, ts, dist):
    """every TU coded with levels uniform over the full int16 range ("stress", what `bench.py --workload idct` times): the two clips of
    the inverse transform (TComTrQuant.cpp:894-948), the saturating residual tiles and the saturating add + clip of the
    motion-compensation epilogue (TComYuv.cpp:264-299) through the production kernels k_itx -> residual tiles -> k_mc_*, not
    through the test-only flat kernel.  "stress" drives ~94 % of the samples into the final clip; "dense" (every position of every TU
    in -3..3) keeps them off it, so that every basis function of every size is visible in the picture."""
    import libhm_amd
    width, height, bd = 1920, 1080, 10
    p = synth.make_picture(width, height, bd, seed=0x484D3136 + int(10 * mode_probs[3]) + int(ts), mode_probs=mode_probs, cbf_prob=1.0,
                           coef_dist=dist, sao=False, tr_split_prob=tr_split, intra_frac=intra, ref_handles=([0], [1]))
    if ts:
        m = dict(p.meta_np)
        rng = np.random.RandomState(9)
        log2tu = 6 - m["depth"] - m["tr_idx"]
        quad = rng.rand(p.num_ctus, 64) < 0.5                                   # one draw per 8x8 area (four z-consecutive partitions)
        m["ts_y"] = ((rng.rand(p.num_ctus, 256) < 0.5) & (log2tu == 2)).astype(np.uint8)
        m["ts_u"] = (np.repeat(quad, 4, axis=1) & (log2tu <= 3)).astype(np.uint8)
        m["ts_v"] = (np.repeat(~quad, 4, axis=1) & (log2tu <= 3)).astype(np.uint8)
        p.meta = abi.MetaHolder(m)
    ref0 = synth.noise_planes(width, height, bd, 11)
    ref1 = synth.blocky_planes(width, height, bd, 12)
    cur = synth.blocky_planes(width, height, bd, 13)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, [p.slice], p.meta, p.coeffs, want, [ref0, ref1])
    with libhm_amd.Context(p.seq) as ctx:
        h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(h1, ref1)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "reconstruction comp %d" % c
        # the same through the batch entry point with compact levels (what a parser that appends TU after TU hands over)
        ctx.upload(hc, cur)
        ctx.decompress_pictures([(hc, [p.slice], p.meta, ctx.pack_levels(p.meta, p.coeffs))])
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "compact levels, comp %d" % c


@pytest.mark.parametrize("flags", [15, 1, 2, 4, 8])
def test_range_extension_residual_tools(oracle, flags):
    """sps_range_extension() in 4:2:0 on the production kernels: rotation of 4x4 intra blocks that skipped the transform, implicit RDPCM
    (intra, modes 10 / 26, with the edge filters off in lossless CUs), explicit RDPCM (inter, mode per TU in bits 1-2 of the
    transform-skip byte) on transform-skip (log2_max_transform_skip_block_size = 5) and cu_transquant_bypass TUs of every size up to
    32x32 (TComTrQuant.cpp:1475-1487, 1737-1792; TComPrediction.cpp:476); intra reference smoothing switched off.  flags: all tools, then each alone (the others must then stay inert although the per-block
    data that would drive them is present)."""
    import libhm_amd
    width, height, bd = 832, 480, 8
    p = synth.make_picture(width, height, bd, seed=0x52457874 + flags, mode_probs=(0.15, 0.25, 0.3, 0.3, 0), cbf_prob=0.9, sao=False,
                           tr_split_prob=0.4, intra_frac=0.4, ref_handles=([0], [1]))
    p.seq.range_ext_flags = flags
    m = dict(p.meta_np)
    rng = np.random.RandomState(flags)
    z = np.arange(256)[None, :]
    cu_first = z & ~((256 >> (2 * m["depth"])) - 1)                             # first partition of the CU a partition belongs to
    per_cu = lambda r: np.take_along_axis(r, cu_first, axis=1)
    log2tu = 6 - m["depth"] - m["tr_idx"]
    m["bypass"] = (per_cu(rng.rand(p.num_ctus, 256)) < 0.3).astype(np.uint8)
    quad = rng.rand(p.num_ctus, 64) < 0.5
    tu_first = z & ~(np.maximum(256 >> (2 * (m["depth"] + m["tr_idx"])), 1) - 1)
    per_tu = lambda r: np.take_along_axis(r, tu_first, axis=1)
    skip = [per_tu(rng.rand(p.num_ctus, 256)) < 0.6, np.where(log2tu <= 3, np.repeat(quad, 4, axis=1), per_tu(rng.rand(p.num_ctus, 256)) < 0.5),
            np.where(log2tu <= 3, np.repeat(~quad, 4, axis=1), per_tu(rng.rand(p.num_ctus, 256)) < 0.5)]
    for c, k in enumerate(("ts_y", "ts_u", "ts_v")):
        ts = (skip[c] & (m["bypass"] == 0)).astype(np.uint8)
        rd = per_tu(rng.randint(0, 3, size=ts.shape)).astype(np.uint8)
        inter_untransformed = (m["pred_mode"] == 0) & ((ts != 0) | (m["bypass"] != 0))
        m[k] = ts | (np.where(inter_untransformed, rd, 0) << 1).astype(np.uint8)
    pick = per_cu(rng.rand(p.num_ctus, 256))
    m["intra_dir_l"] = np.where(pick < 0.3, 10, np.where(pick < 0.6, 26, m["intra_dir_l"])).astype(np.uint8)
    pick = per_cu(rng.rand(p.num_ctus, 256))
    m["intra_dir_c"] = np.where(pick < 0.25, 10, np.where(pick < 0.5, 26, m["intra_dir_c"])).astype(np.uint8)
    p.meta = abi.MetaHolder(m)
    ref0 = synth.noise_planes(width, height, bd, 11)
    ref1 = synth.blocky_planes(width, height, bd, 12)
    cur = synth.blocky_planes(width, height, bd, 13)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, [p.slice], p.meta, p.coeffs, want, [ref0, ref1])
    plain = [a.copy() for a in cur]
    p.seq.range_ext_flags = 0
    oracle.decompress_ctus(p.seq, [p.slice], p.meta, p.coeffs, plain, [ref0, ref1])
    p.seq.range_ext_flags = flags
    assert any(not np.array_equal(a, b) for a, b in zip(want, plain))          # the tools do something on this picture
    with libhm_amd.Context(p.seq) as ctx:
        h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(h1, ref1)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "reconstruction comp %d" % c
        ctx.upload(hc, cur)
        ctx.decompress_pictures([(hc, [p.slice], p.meta, ctx.pack_levels(p.meta, p.coeffs))])
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "compact levels, comp %d" % c


@pytest.mark.parametrize("mode_probs,intra", [((0, 0, 0, 1, 0), 0.0), ((1, 0, 0, 0, 0), 0.0), ((0, 1, 0, 0, 0), 0.3),
                                              ((0, 0, 0.5, 0, 0.5), 0.1), ((0.25, 0.25, 0.25, 0.25, 0), 0.0)])
def test_fused_loop_filter_matches_oracle_on_partition_extremes(oracle, mode_probs, intra):
    """all three loop-filter stages in ONE call (the fused kernel: edge units classified, compacted and applied per 64x64 tile)
    on pictures whose tiles hold every edge unit (all 8x8 CUs: three full waves of units), almost none (64x64 CUs) and mixes"""
    import libhm_amd
    width, height, bd = 1920, 1080, 10
    p = synth.make_picture(width, height, bd, seed=77 + int(100 * intra) + int(10 * mode_probs[3]), intra_frac=intra,
                           mode_probs=mode_probs, ref_handles=([0], [1]))
    ref0 = synth.noise_planes(width, height, bd, 21)
    ref1 = synth.blocky_planes(width, height, bd, 22)
    cur = synth.blocky_planes(width, height, bd, 23)
    _, _, want_fin = _oracle_chain(oracle, p, cur, [ref0, ref1])
    with libhm_amd.Context(p.seq) as ctx:
        h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(h1, ref1)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_fin[c]), "final picture comp %d" % c


@pytest.mark.parametrize("width,height,bd", [(832, 480, 10), (3840, 2160, 10)])
def test_pictures_batch_entry_matches_oracle(oracle, width, height, bd):
    """hmgpu_decompress_pictures + hmgpu_filter_pictures: several independent pictures per call (one batch of launches, inputs staged
    on the copy stream), from ordinary arrays and from a staging block (one DMA for the metadata, one for the levels); called twice
    on the same pictures, as a decoder that reuses its buffers does"""
    import libhm_amd
    n = 3
    pics = [synth.make_picture(width, height, bd, seed=300 + i, bi=(i == 1), intra_frac=0.1 * i, ref_handles=([0], [1])) for i in range(n)]
    ref0 = synth.noise_planes(width, height, bd, 41)
    ref1 = synth.blocky_planes(width, height, bd, 42)
    cur = synth.blocky_planes(width, height, bd, 43)
    want = [_oracle_chain(oracle, p, cur, [ref0, ref1])[2] for p in pics]
    with libhm_amd.Context(abi.make_seq(width, height, bd, bd, log2_ctu=6, max_pictures=2 + n)) as ctx:
        h0, h1 = ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(h1, ref1)
        hs = [ctx.acquire() for _ in range(n)]
        stg = ctx.staging_alloc()
        packed1 = ctx.pack_levels(pics[1].meta, pics[1].coeffs)      # picture 1: compact levels (coded TUs only) from ordinary memory
        sao = [abi.sao_array_from_raw(p.sao_raw) for p in pics]
        assert 0 < int(packed1.starts[0][-1]) < pics[1].coeffs.arrays[0].size
        for rnd in range(2):
            for h in hs:
                ctx.upload(h, cur)
            # picture 2 travels through a staging block: dense levels in the first round, compact ones in the second
            if rnd == 0:
                stg.fill(pics[2].meta, pics[2].coeffs)
            else:
                ctx.sync()
                assert stg.fill_compact(libhm_amd.lib(), ctx.seq, pics[2].meta, pics[2].coeffs) < 2 * sum(a.size for a in pics[2].coeffs.arrays)
            stg.set_groups(intra=True, flags=(rnd == 0))     # second round: the flag arrays (all zero here) stay at home
            ctx.decompress_pictures([(hs[0], [pics[0].slice], pics[0].meta, pics[0].coeffs), (hs[1], [pics[1].slice], pics[1].meta, packed1),
                                     (hs[2], [pics[2].slice], stg, stg)])
            ctx.filter_pictures([(hs[i], pics[i].pp, sao[i]) for i in range(n)])
            for i in range(n):
                got = ctx.download(hs[i])
                for c in range(3):
                    assert np.array_equal(got[c], want[i][c]), "round %d picture %d comp %d" % (rnd, i, c)
        # a finished picture of the batch is a reference like any other
        q = synth.make_picture(width, height, bd, seed=310, ref_handles=([0], [0]))
        _, _, want_q = _oracle_chain(oracle, q, cur, [want[0]])
        qsl = abi.clone_slice(q.slice)
        qsl.ref_pic[0][0] = hs[0]
        hq = h1                                                      # (reuse a buffer nobody references any more)
        ctx.upload(hq, cur)
        ctx.decompress_pictures([(hq, [qsl], q.meta, q.coeffs)])
        ctx.filter_pictures([(hq, q.pp, abi.sao_array_from_raw(q.sao_raw))])
        got = ctx.download(hq)
        for c in range(3):
            assert np.array_equal(got[c], want_q[c]), "picture predicted from a picture of the batch, comp %d" % c
        ctx.staging_free(stg)
        with pytest.raises(libhm_amd.HmgpuError):                    # a picture and its reference in one call: not independent
            ctx.decompress_pictures([(hs[0], [pics[0].slice], pics[0].meta, pics[0].coeffs), (hs[1], [qsl], q.meta, q.coeffs)])


@pytest.mark.parametrize("width,height,bd,cip,one_i_picture", [(832, 480, 10, 0, False), (832, 480, 8, 1, False), (1920, 1080, 10, 0, True),
                                                             (200, 136, 8, 0, False)])
def test_scattered_intra_cus_in_batches(oracle, width, height, bd, cip, one_i_picture):
    """P pictures with scattered intra CUs, five per call: the calls k_intra<1, LEAN> serves (no I slice, less than half of the partitions
    intra: one wave per CTU, reference samples and residual straight from the picture) -- and, with one all-intra I picture among
    them, the general kernel, whose CTUs with few intra areas take the same unstaged path beside staged dense ones.  3 % .. 45 %
    intra CUs, all TU sizes, multi-slice pictures, constrained intra prediction (availability with holes), partial CTUs."""
    import libhm_amd
    fr = [0.03, 0.1, 0.25, 0.45, 0.08]
    n = len(fr)
    pics = []
    for i in range(n):
        probs = [(0.1, 0.3, 0.3, 0.2, 0.1), (0, 0, 0.3, 0.7, 0), (0.3, 0.3, 0.2, 0.1, 0.1), (0, 0.2, 0.4, 0.4, 0), (0, 0, 0, 1, 0)][i]
        if width % 16 or height % 16:
            probs = (0, 0, 0, 1, 0)
        pics.append(synth.make_picture(width, height, bd, seed=0x5CA7 + 7 * i + cip, intra_frac=fr[i], mode_probs=probs, tr_split_prob=0.3 + 0.15 * i,
                                       cbf_prob=0.6, ref_handles=([0], [0]), num_slices=(3 if i == 2 else 1)))
    if one_i_picture:
        pics[3] = synth.make_picture(width, height, bd, seed=0x5CA7 + 99, intra_frac=1.0, cbf_prob=0.6, ref_handles=([0], [0]))
        for sl in pics[3].slices:
            sl.slice_type = abi.I_SLICE
    for p in pics:
        for sl in p.slices:
            sl.constrained_intra_pred = cip
    ref0 = synth.noise_planes(width, height, bd, 51)
    cur = synth.blocky_planes(width, height, bd, 53)
    want_rec, want_fin = [], []
    for p in pics:
        rec = [a.copy() for a in cur]
        oracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, rec, [ref0])
        want_rec.append([a.copy() for a in rec])
        oracle.loop_filter_pic(p.seq, p.slices, p.meta, p.pp, rec, 3)
        prm = oracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
        want_fin.append(oracle.sao_process(p.seq, p.slices, p.pp, p.meta, prm, rec))
    with libhm_amd.Context(abi.make_seq(width, height, bd, bd, log2_ctu=6, max_pictures=1 + n)) as ctx:
        h0 = ctx.acquire()
        ctx.upload(h0, ref0)
        hs = [ctx.acquire() for _ in range(n)]
        for rnd in range(2):                                         # second round: compact levels
            for h in hs:
                ctx.upload(h, cur)
            ctx.decompress_pictures([(hs[i], pics[i].slices, pics[i].meta, pics[i].coeffs if rnd == 0 else ctx.pack_levels(pics[i].meta, pics[i].coeffs))
                                     for i in range(n)])
            for i in range(n):
                got = ctx.download(hs[i])
                for c in range(3):
                    assert np.array_equal(got[c], want_rec[i][c]), "round %d reconstruction of picture %d comp %d" % (rnd, i, c)
            ctx.filter_pictures([(hs[i], pics[i].pp, abi.sao_array_from_raw(pics[i].sao_raw)) for i in range(n)])
            for i in range(n):
                got = ctx.download(hs[i])
                for c in range(3):
                    assert np.array_equal(got[c], want_fin[i][c]), "round %d picture %d comp %d" % (rnd, i, c)


@pytest.mark.parametrize("width,height,bd,slices", [(1920, 1080, 10, 1), (1920, 1080, 8, 1), (832, 480, 10, 4)])
def test_custom_scaling_lists_on_every_list_id(oracle, width, height, bd, slices):
    """scaling lists other than HM's defaults on every list id -- intra and inter, luma / Cb / Cr, 4x4 .. 32x32 with their DC entries -- on a
    picture with intra and inter CUs and every TU size: de-quantisation with m_dequantCoef (TComTrQuant.cpp:1203-1313, setScalingListDec /
    processScalingListDec :1920-1959) through k_itx, for the inter residual and for the residual k_intra adds; one picture in four slices
    (all slices of a picture name the same PPS, hence the same lists).  (The HM-made scaling-list streams are 208x120.)"""
    import ctypes
    import libhm_amd
    p = synth.make_picture(width, height, bd, seed=0x5CA1 + bd + slices, intra_frac=0.4, cbf_prob=0.85, tr_split_prob=0.5, sao=False,
                           mode_probs=(0.15, 0.25, 0.3, 0.2, 0.1), ref_handles=([0], [0]), num_slices=slices)
    rng = np.random.RandomState(0x11575 + bd)
    keep = []
    for k in range(1):
        lists = abi.ScalingLists()
        for sz in range(4):
            for l in range(6):
                lists.dc[sz][l] = int(rng.randint(1, 256)) if sz >= 2 else 16
                n = 16 if sz == 0 else 64
                # a ramp as encoders send it, a flat one and a wild one, list by list
                kind = (sz + l + k) % 3
                for i in range(64):
                    v = 16
                    if i < n:
                        v = int(rng.randint(1, 256)) if kind == 0 else (min(255, 8 + 3 * i + l) if kind == 1 else 16 + 8 * l)
                    lists.coef[sz][l][i] = v
        keep.append(lists)
    for sl in p.slices:
        sl.scaling_lists = ctypes.pointer(keep[0])
    ref0 = synth.noise_planes(width, height, bd, 61)
    cur = synth.blocky_planes(width, height, bd, 63)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, want, [ref0])
    # (the lists matter: the same picture with HM's flat default differs)
    flat = [a.copy() for a in cur]
    plain = [abi.clone_slice(sl) for sl in p.slices]
    for sl in plain:
        sl.scaling_lists = None
    oracle.decompress_ctus(p.seq, plain, p.meta, p.coeffs, flat, [ref0])
    assert any(not np.array_equal(flat[c], want[c]) for c in range(3))
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc = ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref0)
        ctx.upload(hc, cur)
        ctx.decompress_pictures([(hc, p.slices, p.meta, p.coeffs)])
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "reconstruction comp %d" % c
        ctx.upload(hc, cur)
        ctx.decompress_pictures([(hc, p.slices, p.meta, ctx.pack_levels(p.meta, p.coeffs))])
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "compact levels, comp %d" % c


@pytest.mark.parametrize("across", [0, 1])
def test_multi_slice_picture_matches_oracle(oracle, across):
    """five slices starting at arbitrary CTUs (own QP / deblocking offsets), one hmgpu_decompress_slice call per slice: slice
    borders bound intra references, deblocking (slice_loop_filter_across_slices_enabled_flag) and SAO neighbourhoods"""
    import libhm_amd
    w, h, bd = 832, 480, 10
    p = synth.make_picture(w, h, bd, seed=91 + across, intra_frac=0.3, ref_handles=([0], [0]), num_slices=5, lf_across_slices=across)
    assert len(p.slices) == 5 and any(a % p.ctus_w for a, _ in p.slice_ranges)
    ref = synth.noise_planes(w, h, bd, 5)
    cur = synth.blocky_planes(w, h, bd, 6)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, want, [ref])
    want_rec = [a.copy() for a in want]
    oracle.loop_filter_pic(p.seq, p.slices, p.meta, p.pp, want, 3)
    prm = oracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
    want_fin = oracle.sao_process(p.seq, p.slices, p.pp, p.meta, prm, want)
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc = ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref)
        ctx.upload(hc, cur)
        for k, (first, num) in enumerate(p.slice_ranges):
            ctx.decompress_slice(hc, k, p.slices[k], p.meta, p.coeffs, first_ctu=first, num_ctus=num)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_rec[c]), "reconstruction comp %d" % c
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_fin[c]), "filtered comp %d" % c
        # the same picture again as a batched replay (five slice calls per picture)
        ctx.replay([hc], 15, 1)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], want_fin[c]), "replayed comp %d" % c


def test_identical_motion_collapses_to_uni(oracle):
    """B slice, both lists point at the same POC with the same MV: xCheckIdenticalMotion -> uni-prediction rounding"""
    import libhm_amd
    w, h, bd = 256, 128, 10
    p = synth.make_picture(w, h, bd, seed=5, bi=True, ref_handles=([0], [0]))
    p.slice.ref_poc[1][0] = p.slice.ref_poc[0][0]
    m = dict(p.meta_np)
    m["mv1"] = np.where((m["ref_idx1"] >= 0)[:, :, None], m["mv0"], 0)            # same MV wherever both lists are used
    both = (m["ref_idx0"] >= 0) & (m["ref_idx1"] >= 0)
    m["mv1"] = np.where(both[:, :, None], m["mv0"], m["mv1"])
    p.meta = abi.MetaHolder(m)
    ref = synth.noise_planes(w, h, bd, 3)
    cur = synth.noise_planes(w, h, bd, 4)
    want = [a.copy() for a in cur]
    oracle.decompress_ctus(p.seq, [p.slice], p.meta, p.coeffs, want, [ref])
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc = ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref)
        ctx.upload(hc, cur)
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        got = ctx.download(hc)
    for c in range(3):
        assert np.array_equal(got[c], want[c])


def test_properties_at_2160p(oracle):
    """size-independent properties: (1) zero motion + no residual reproduces the reference picture; (2) SAO with all
    offsets zero and deblocking disabled leave the picture untouched; (3) a batched replay of independent pictures gives
    the same samples as picture-by-picture calls; (4) so does the two-lane replay -- and those samples are the oracle's."""
    import libhm_amd
    w, h, bd = 3840, 2160, 10
    p = synth.make_picture(w, h, bd, seed=21, cbf_prob=0.0, mv_range=0, ref_handles=([0], [0]))
    m = dict(p.meta_np)
    m["mv0"] = np.zeros_like(m["mv0"])
    p.meta = abi.MetaHolder(m)
    p.sao_raw[:, :, 3:] = 0
    p.slice.deblocking_disable = 1
    ref = synth.noise_planes(w, h, bd, 31)
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc, hd = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.upload(h0, ref)
        ctx.upload(hc, synth.noise_planes(w, h, bd, 32))
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        got = ctx.download(hc)
        for c in range(3):
            assert np.array_equal(got[c], ref[c])
        # (3) two independent pictures with real work, replayed as one batch
        q = synth.make_picture(w, h, bd, seed=22, ref_handles=([0], [0]))
        _, _, want_q = _oracle_chain(oracle, q, synth.noise_planes(w, h, bd, 33), [ref])
        for pic in (hc, hd):
            ctx.upload(pic, synth.noise_planes(w, h, bd, 33))
            ctx.decompress_slice(pic, 0, q.slice, q.meta, q.coeffs)
            ctx.filter_picture(pic, q.pp, q.sao_raw)
        single = ctx.download(hc)
        digest_single = gu.hm_md5(single, [bd] * 3)
        ctx.replay([hc, hd], 15, 2)
        a, b = ctx.download(hc), ctx.download(hd)
        assert gu.hm_md5(a, [bd] * 3) == digest_single
        assert gu.hm_md5(b, [bd] * 3) == digest_single
        # (4) the same batch as two lanes on two streams (hmgpu_set_streams): the pictures do not change
        ctx.set_streams(2)
        ctx.replay([hc, hd], 15, 3)
        a, b = ctx.download(hc), ctx.download(hd)
        for c in range(3):
            assert np.array_equal(a[c], want_q[c]) and np.array_equal(b[c], want_q[c]), "two-lane replay vs oracle, comp %d" % c
        ctx.set_streams(1)
        with pytest.raises(libhm_amd.HmgpuError):
            ctx.set_streams(3)


def test_unsupported_tools_are_refused():
    import libhm_amd
    p = synth.make_picture(128, 64, 8, seed=2, ref_handles=([0], [0]))
    with libhm_amd.Context(p.seq) as ctx:
        h0, hc = ctx.acquire(), ctx.acquire()
        m = dict(p.meta_np)
        m["ipcm"] = np.ones_like(m["depth"])
        with pytest.raises(libhm_amd.HmgpuError) as e:                     # PCM CUs without their sample buffers
            ctx.decompress_slice(hc, 0, p.slice, abi.MetaHolder(m), p.coeffs)
        assert e.value.status == abi.HMGPU_EINVAL
        with pytest.raises(libhm_amd.HmgpuError) as e:                     # reference handle that is not a live picture
            p.slice.ref_pic[0][0] = 7
            ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        assert e.value.status == abi.HMGPU_EINVAL
        assert h0 == 0
