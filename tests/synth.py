"""Synthetic HM-shaped pictures for full-size parity tests and for bench.py.

Produces exactly what HM's parser would leave behind for a picture -- per-CTU TComDataCU arrays in z-scan order,
coefficient levels in HM's layout, raw SAO parameters -- but drawn from a seeded PRNG instead of a bitstream:
  * CTU partitioning drawn from {64x64, 4x32x32, 16x16x16, 64x8x8, AMP at 32x32} with p = {.1,.3,.3,.2,.1}
  * one motion vector per PU: integer part uniform in [-64, 64] luma samples, fraction uniform over the 16 phases
  * transform trees of depth 0/1, cbf with p = 0.5 per TU and component, "typical" levels: top-left 8x8 (4x4)
    Laplacian(b = 12) with P(nonzero) = 0.35
  * QP uniform in [22, 37] per CTU; optional fraction of intra CUs (Bs = 2 edges for the loop filter)
  * SAO per CTU and component: off .3 / BO .2 / EO0-3 .125 each
Everything is numpy-vectorised so that a 3840x2160 picture takes about a second to draw.
"""
import numpy as np

from libhm_amd import abi


def _zxy(parts):
    z = np.arange(parts)
    x = np.zeros(parts, dtype=np.int64)
    y = np.zeros(parts, dtype=np.int64)
    for b in range(8):
        x |= ((z >> (2 * b)) & 1) << b
        y |= ((z >> (2 * b + 1)) & 1) << b
    return x, y


class SynthPicture:
    pass


def make_picture(width, height, bit_depth=10, seed=1, bi=False, intra_frac=0.0, num_refs=1, slice_qp_range=(22, 37),
                 cbf_prob=0.5, sao=True, mode_probs=(0.1, 0.3, 0.3, 0.2, 0.1), ref_handles=None, mv_range=64,
                 coef_dist="typical", tr_split_prob=0.35, intra_modes=True, num_slices=1, lf_across_slices=1):
    """Returns a SynthPicture with .seq, .slice (abi.SliceParams), .meta (MetaHolder), .coeffs (CoeffHolder),
    .sao_raw [num_ctus,3,35], .pp, .meta_np.  ref_handles: device picture handles of list-0 / list-1 references.
    coef_dist: "typical" (see above), "stress" (every level of a coded TU uniform over the full int16 range) or
    "dense" (every level uniform in -3..3)."""
    rng = np.random.RandomState(seed)
    ctu, pw, parts = 64, 16, 256
    cw, ch = (width + 63) // 64, (height + 63) // 64
    n = cw * ch
    zx, zy = _zxy(parts)
    ctu_x = (np.arange(n) % cw) * 64
    ctu_y = (np.arange(n) // cw) * 64
    px = ctu_x[:, None] + 4 * zx[None, :]
    py = ctu_y[:, None] + 4 * zy[None, :]
    inside = (px < width) & (py < height)

    # ---- CTU partitioning mode.  CTUs cut by the picture border take the modes whose CUs fit (16x16 / 8x8)
    mode = rng.choice(5, size=n, p=mode_probs)
    partial = (ctu_x + 64 > width) | (ctu_y + 64 > height)
    if width % 16 or height % 16:
        mode[partial] = 3                                          # only 8x8 CUs tile a picture that is not a multiple of 16
    else:
        mode[partial] = np.where(rng.rand(int(partial.sum())) < 0.6, 2, 3)
    cu_log2 = np.array([6, 5, 4, 3, 5])[mode]                     # per CTU
    depth = (6 - cu_log2)[:, None] * np.ones((1, parts), dtype=np.int64)
    cu_parts = (1 << (2 * (cu_log2 - 2)))[:, None]                # partitions per CU
    z = np.arange(parts)[None, :]
    cu_idx = z // cu_parts                                         # CU index inside the CTU (z order)
    child = (z % cu_parts) // np.maximum(cu_parts // 4, 1)         # quadrant inside the CU
    n_cu_max = 64

    # ---- part size: 2Nx2N except mode 4 (AMP, one of 2NxnU, 2NxnD, nLx2N, nRx2N per CU)
    amp_type = rng.randint(4, 8, size=(n, n_cu_max))
    part_size = np.zeros((n, parts), dtype=np.int64)
    is_amp = (mode == 4)[:, None] & np.ones((1, parts), dtype=bool)
    part_size[is_amp] = np.take_along_axis(amp_type, cu_idx, axis=1)[is_amp]
    # PU index inside the CU (0/1) for AMP CUs: depends on the partition's row/column inside the 32x32 CU (8 partitions wide)
    rel_x = zx[None, :] % 8
    rel_y = zy[None, :] % 8
    pu_in_cu = np.zeros((n, parts), dtype=np.int64)
    pu_in_cu = np.where(is_amp & (part_size == abi.SIZE_2NxnU), (rel_y >= 2).astype(np.int64), pu_in_cu)
    pu_in_cu = np.where(is_amp & (part_size == abi.SIZE_2NxnD), (rel_y >= 6).astype(np.int64), pu_in_cu)
    pu_in_cu = np.where(is_amp & (part_size == abi.SIZE_nLx2N), (rel_x >= 2).astype(np.int64), pu_in_cu)
    pu_in_cu = np.where(is_amp & (part_size == abi.SIZE_nRx2N), (rel_x >= 6).astype(np.int64), pu_in_cu)
    pu_idx = cu_idx * 2 + pu_in_cu                                 # < 128

    # ---- prediction mode per CU
    intra_cu = rng.rand(n, n_cu_max) < intra_frac
    intra = np.take_along_axis(intra_cu, cu_idx, axis=1)
    intra &= ~is_amp                                               # AMP is inter only
    pred_mode = intra.astype(np.int64)

    # ---- motion per PU
    def draw_mv():
        mvi = rng.randint(-mv_range, mv_range + 1, size=(n, 128, 2))
        mvf = rng.randint(0, 4, size=(n, 128, 2))
        return mvi * 4 + mvf
    mv0_pu = draw_mv()
    mv1_pu = draw_mv()
    idx3 = np.repeat(pu_idx[:, :, None], 2, axis=2)
    mv0 = np.take_along_axis(mv0_pu, idx3, axis=1)
    mv1 = np.take_along_axis(mv1_pu, idx3, axis=1)
    ref_idx0 = np.zeros((n, parts), dtype=np.int64)
    if num_refs > 1:
        ref_pu = rng.randint(0, num_refs, size=(n, 128))
        ref_idx0 = np.take_along_axis(ref_pu, pu_idx, axis=1)
    ref_idx1 = np.full((n, parts), -1, dtype=np.int64)
    if bi:
        # per PU: 0 = L0 only, 1 = L1 only, 2 = both   (8x4 / 4x8 PUs do not exist here, so bi is legal everywhere)
        kind_pu = rng.choice(3, size=(n, 128), p=(0.2, 0.1, 0.7))
        kind = np.take_along_axis(kind_pu, pu_idx, axis=1)
        ref_idx1 = np.where(kind >= 1, 0, -1)
        ref_idx0 = np.where(kind == 1, -1, ref_idx0)
    ref_idx0 = np.where(intra, -1, ref_idx0)
    ref_idx1 = np.where(intra, -1, ref_idx1)
    mv0 = np.where((ref_idx0 < 0)[:, :, None], 0, mv0)
    mv1 = np.where((ref_idx1 < 0)[:, :, None], 0, mv1)

    # ---- transform tree: tr_idx in {0,1} per CU (64x64 CUs and AMP CUs always split once)
    tr_cu = (rng.rand(n, n_cu_max) < tr_split_prob).astype(np.int64)
    tr_idx = np.take_along_axis(tr_cu, cu_idx, axis=1)
    tr_idx = np.where(((mode == 0) | (mode == 4))[:, None], 1, tr_idx)
    # cbf: per CU one flag for the unsplit TU and four for the children, per component
    cbf = []
    for comp in range(3):
        c0 = rng.rand(n, n_cu_max) < cbf_prob
        c1 = rng.rand(n, n_cu_max, 4) < cbf_prob
        if comp > 0:
            # 8x8 CUs split once: a single 4x4 chroma TU for the four 4x4 luma TUs; its flag is stored at both depths
            one = np.repeat(c1[:, :, :1], 4, axis=2)
            c1 = np.where((mode == 3)[:, None, None], one, c1)
        any1 = c1.any(axis=2)
        b0_unsplit = np.take_along_axis(c0, cu_idx, axis=1)
        b0_split = np.take_along_axis(any1, cu_idx, axis=1)
        flat = c1.reshape(n, n_cu_max * 4)
        b1 = np.take_along_axis(flat, cu_idx * 4 + child, axis=1)
        bits = np.where(tr_idx == 0, b0_unsplit.astype(np.int64), b0_split.astype(np.int64) | (b1.astype(np.int64) << 1))
        cbf.append(bits)
    # ---- intra prediction modes: one luma mode per CU (2Nx2N), chroma mode from HM's candidate set incl. DM (36)
    luma_mode_cu = rng.randint(0, 35, size=(n, n_cu_max))
    intra_dir_l = np.take_along_axis(luma_mode_cu, cu_idx, axis=1)
    chroma_mode_cu = np.array([0, 1, 10, 26, 34, 36])[rng.randint(0, 6, size=(n, n_cu_max))]
    intra_dir_c = np.take_along_axis(chroma_mode_cu, cu_idx, axis=1)
    qp_ctu = rng.randint(slice_qp_range[0], slice_qp_range[1] + 1, size=n)
    qp = np.repeat(qp_ctu[:, None], parts, axis=1)

    # undecoded partitions (outside the picture): HM's initCU defaults
    def outside(a, v):
        return np.where(inside, a, v)
    part_size = outside(part_size, abi.SIZE_NONE)
    depth = outside(depth, 0)
    pred_mode = outside(pred_mode, 0)
    tr_idx = outside(tr_idx, 0)
    cbf = [outside(c, 0) for c in cbf]
    ref_idx0 = outside(ref_idx0, -1)
    ref_idx1 = outside(ref_idx1, -1)
    mv0 = np.where(inside[:, :, None], mv0, 0)
    mv1 = np.where(inside[:, :, None], mv1, 0)

    # ---- coefficient levels in HM's layout
    coef = [np.zeros((n, 4096), dtype=np.int16), np.zeros((n, 1024), dtype=np.int16), np.zeros((n, 1024), dtype=np.int16)]
    log2cu_p = (6 - depth)
    log2tu_p = log2cu_p - tr_idx
    decoded = inside & (part_size != abi.SIZE_NONE)
    for comp in range(3):
        chain = (1 << (tr_idx + 1)) - 1
        has = decoded & ((cbf[comp] & chain) == chain)
        for log2tu_l in range(2, 6):           # luma TU size of the node
            tu_parts = 1 << (2 * max(log2tu_l - 2, 0))
            sel = has & (log2tu_p == log2tu_l)
            if comp == 0 or log2tu_l > 2:
                origin = sel & ((np.arange(parts)[None, :] % tu_parts) == 0)
                size = (1 << log2tu_l) >> (1 if comp else 0)
            else:
                origin = sel & ((np.arange(parts)[None, :] % 4) == 0)      # shared 4x4 chroma TU rides with the first child
                size = 4
            a_idx, z_idx = np.nonzero(origin)
            if a_idx.size == 0:
                continue
            if coef_dist == "stress":
                k = size
                lev = rng.randint(-32768, 32768, size=(a_idx.size, k, k)).astype(np.int16)
            elif coef_dist == "dense":       # every position of the TU small and non-zero-ish: all basis functions, few samples reach the final clip
                k = size
                lev = rng.randint(-3, 4, size=(a_idx.size, k, k)).astype(np.int16)
            else:
                k = min(size, 8)
                lev = np.round(rng.laplace(0, 12, size=(a_idx.size, k, k))) * (rng.rand(a_idx.size, k, k) < 0.35)
                lev = np.clip(lev, -32768, 32767).astype(np.int16)
            off = (16 if comp == 0 else 4) * z_idx
            rr, cc = np.meshgrid(np.arange(k), np.arange(k), indexing="ij")
            flat_idx = off[:, None, None] + rr[None] * size + cc[None]
            coef[comp][a_idx[:, None, None], flat_idx] = lev

    # ---- SAO
    sao_raw = np.zeros((n, 3, 35), dtype=np.int32)
    if sao:
        maxo = 7
        for comp in range(3):
            kind = rng.choice(6, size=n, p=(0.3, 0.2, 0.125, 0.125, 0.125, 0.125))     # 0 off, 1 BO, 2.. EO0..3
            sao_raw[:, comp, 0] = np.where(kind == 0, abi.SAO_OFF, abi.SAO_NEW)
            sao_raw[:, comp, 1] = np.where(kind == 1, abi.SAO_BO, np.maximum(kind - 2, 0))
            band = rng.randint(0, 32, size=n)
            sao_raw[:, comp, 2] = np.where(kind == 1, band, 0)
            offs = rng.randint(-maxo, maxo + 1, size=(n, 4))
            for i in range(4):
                bidx = (band + i) % 32
                bo = np.zeros((n, 32), dtype=np.int32)
                bo[np.arange(n), bidx] = offs[:, i]
                sao_raw[:, comp, 3:] += np.where((kind == 1)[:, None], bo, 0)
            eo = np.zeros((n, 32), dtype=np.int32)
            eo[:, 0] = np.abs(offs[:, 0]); eo[:, 1] = np.abs(offs[:, 1]); eo[:, 3] = -np.abs(offs[:, 2]); eo[:, 4] = -np.abs(offs[:, 3])
            sao_raw[:, comp, 3:] += np.where((kind >= 2)[:, None], eo, 0)

    p = SynthPicture()
    p.width, p.height, p.bit_depth, p.num_ctus, p.ctus_w = width, height, bit_depth, n, cw
    p.seq = abi.make_seq(width, height, bit_depth, bit_depth, log2_ctu=6, max_pictures=4)
    handles = ref_handles if ref_handles is not None else ([0] * num_refs, [0])
    l0 = list(handles[0])[:max(num_refs, 1)]
    l1 = list(handles[1])[:1] if bi else []
    p.slice = abi.make_slice(abi.B_SLICE if bi else abi.P_SLICE, (l0, l1), ([100 + i for i in range(len(l0))], [200 + i for i in range(len(l1))]))
    # slices: contiguous CTU ranges starting at seeded CTU addresses (mid-row starts included); slice k may differ in its
    # deblocking offsets and chroma QP offsets, all share the reference lists
    p.slices, p.slice_ranges = [p.slice], [(0, n)]
    slice_idx = np.zeros(n, dtype=np.uint16)
    if num_slices > 1:
        starts = [0] + sorted(int(v) for v in rng.choice(np.arange(1, n), size=num_slices - 1, replace=False))
        p.slices, p.slice_ranges = [], []
        for k, a in enumerate(starts):
            b_ = starts[k + 1] if k + 1 < len(starts) else n
            sl = abi.make_slice(abi.B_SLICE if bi else abi.P_SLICE, (l0, l1), ([100 + i for i in range(len(l0))], [200 + i for i in range(len(l1))]),
                                cb_qp_offset=int(rng.randint(-3, 4)), cr_qp_offset=int(rng.randint(-3, 4)),
                                beta_offset_div2=int(rng.randint(-2, 3)), tc_offset_div2=int(rng.randint(-2, 3)),
                                lf_across_slices=lf_across_slices)
            sl.pps_cb_qp_offset, sl.pps_cr_qp_offset = sl.cb_qp_offset, sl.cr_qp_offset
            p.slices.append(sl)
            p.slice_ranges.append((a, b_ - a))
            slice_idx[a:b_] = k
        p.slice = p.slices[0]
    m = {"depth": depth, "part_size": part_size, "pred_mode": pred_mode, "qp": qp, "tr_idx": tr_idx, "cbf_y": cbf[0], "cbf_u": cbf[1],
         "cbf_v": cbf[2], "mv0": mv0, "mv1": mv1, "ref_idx0": ref_idx0, "ref_idx1": ref_idx1,
         "intra_dir_l": np.where(intra, intra_dir_l, 1), "intra_dir_c": np.where(intra, intra_dir_c, 36)}
    m["slice_idx"] = slice_idx
    if not intra_modes:                 # the caller leaves intra CUs to somebody else: no modes, nothing reconstructed there
        del m["intra_dir_l"], m["intra_dir_c"]
    p.meta_np = m
    p.meta = abi.MetaHolder(m)
    p.coeffs = abi.CoeffHolder(*coef)
    p.sao_raw = sao_raw
    p.pp = abi.make_pic_params(sao_enabled=1 if sao else 0)
    p.inside = inside
    p.intra = intra & decoded
    p.px, p.py = px, py
    return p


def intra_sample_mask(p, comp):
    """True where a sample belongs to an intra CU (not reconstructed by the device in this round)"""
    cs = 1 if comp else 0
    mask = np.zeros((p.height >> cs, p.width >> cs), dtype=bool)
    a, z = np.nonzero(p.intra)
    if a.size:
        x, y = p.px[a, z] >> cs, p.py[a, z] >> cs
        step = 4 >> cs
        for dy in range(step):
            for dx in range(step):
                mask[y + dy, x + dx] = True
    return mask


def noise_planes(width, height, bit_depth, seed):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 1 << bit_depth, size=(height >> (1 if c else 0), width >> (1 if c else 0))).astype(np.int16) for c in range(3)]


def blocky_planes(width, height, bit_depth, seed):
    """low-pass noise + 8x8 blocking steps: splits the deblocking decisions between off / weak / strong"""
    rng = np.random.RandomState(seed)
    out = []
    for c in range(3):
        w, h = width >> (1 if c else 0), height >> (1 if c else 0)
        base = rng.randint(0, 1 << bit_depth, size=((h + 15) // 16 + 1, (w + 15) // 16 + 1)).astype(np.float64)
        up = np.kron(base, np.ones((16, 16)))[:h, :w]
        for _ in range(2):
            up = (up + np.roll(up, 3, 0) + np.roll(up, 3, 1) + np.roll(up, -3, 0) + np.roll(up, -3, 1)) / 5.0
        steps = rng.randint(-6, 7, size=((h + 7) // 8, (w + 7) // 8)) * (1 << (bit_depth - 8))
        up = up * 0.6 + (1 << (bit_depth - 1)) * 0.4 + np.kron(steps, np.ones((8, 8)))[:h, :w] + rng.randint(-2, 3, size=(h, w))
        out.append(np.clip(np.round(up), 0, (1 << bit_depth) - 1).astype(np.int16))
    return out
