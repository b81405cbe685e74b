// filter_core.h -- device functions shared by the stand-alone loop-filter kernels (k_dbk.hip, k_sao.hip) and the fused one
// (k_filter.hip): boundary strength, luma / chroma deblocking of one edge unit (TComLoopFilter.cpp:411-891), SAO offset
// arithmetic (TComSampleAdaptiveOffset.cpp:404-631).
#pragma once
#include "hmgpu_dev.h"

namespace hmgpu {

static __constant__ uint8_t c_tc_table[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
static __constant__ uint8_t c_beta_table[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                         16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54,
                                         56, 58, 60, 62, 64};
static __constant__ uint8_t c_chroma_scale_420_dbk[58] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};

__device__ inline bool mvd4(const int16_t* a, const int16_t* b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }
// the same on packed vectors (hor | ver << 16): some component differs by 4 or more quarter samples.  The differences of two 16-bit
// vector components fit 17 bits: saturating subtraction keeps the comparison exact (|d| >= 4 survives the clamp)
typedef short mv2 __attribute__((ext_vector_type(2)));
__device__ inline bool mvd4(uint32_t a, uint32_t b) {
  const mv2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(mv2, a), __builtin_bit_cast(mv2, b));
  const mv2 m = __builtin_elementwise_max(d, __builtin_elementwise_sub_sat((mv2){0, 0}, d));       // |d| (saturated)
  return (__builtin_bit_cast(uint32_t, m) & 0xfffcfffcu) != 0u;
}

// xGetBoundaryStrengthSingle (TComLoopFilter.cpp:411-537).  `transform_edge` is the m_aapucBS marker.  The P-slice
// branch (:515-532) is the B-slice branch with list 1 absent on both sides, so one formula serves both.
__device__ inline int boundary_strength(const BlkInfo& p, const BlkInfo& q, bool transform_edge) {
  if ((p.flags | q.flags) & BF_INTRA) return 2;
  if (transform_edge && ((p.flags | q.flags) & BF_CBFY)) return 1;
  const int p0 = p.ref[0], p1 = p.ref[1], q0 = q.ref[0], q1 = q.ref[1];
  if ((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0)) {
    const uint32_t pm0 = __builtin_bit_cast(uint32_t, *reinterpret_cast<const mv2*>(p.mv[0])), pm1 = __builtin_bit_cast(uint32_t, *reinterpret_cast<const mv2*>(p.mv[1]));
    const uint32_t qm0 = __builtin_bit_cast(uint32_t, *reinterpret_cast<const mv2*>(q.mv[0])), qm1 = __builtin_bit_cast(uint32_t, *reinterpret_cast<const mv2*>(q.mv[1]));
    const bool straight = mvd4(qm0, pm0) || mvd4(qm1, pm1), crossed = mvd4(qm1, pm0) || mvd4(qm0, pm1);
    if (p0 != p1) return (p0 == q0 ? straight : crossed) ? 1 : 0;
    return (straight && crossed) ? 1 : 0;
  }
  return 1;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ inline s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ inline u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ inline uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline s16x2 splat(int v) { return (s16x2){(short)v, (short)v}; }
__device__ inline u16x2 usplat(int v) { return (u16x2){(unsigned short)v, (unsigned short)v}; }
__device__ inline s16x2 pk_clip(s16x2 lo, s16x2 hi, s16x2 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }

// Two lines across a luma edge at once (xPelFilterLuma, :800-859): s[k] holds sample k (p3 p2 p1 p0 q0 q1 q2 q3) of the two
// lines in its 16-bit halves.  Samples are at most 10 bits wide (hmgpu_set_sequence refuses more), so every intermediate of
// the HM formulas -- the largest is 9*(q0-p0) - 3*(q1-p1) + 8, |.| <= 12284 -- is exact in 16 bits.
__device__ inline void filter_luma_pair_strong(uint32_t (&s)[8], int tc) {
  const u16x2 m0 = as_u16x2(s[0]), m1 = as_u16x2(s[1]), m2 = as_u16x2(s[2]), m3 = as_u16x2(s[3]);
  const u16x2 m4 = as_u16x2(s[4]), m5 = as_u16x2(s[5]), m6 = as_u16x2(s[6]), m7 = as_u16x2(s[7]);
  const s16x2 t2 = splat(2 * tc);
  const u16x2 s23 = m2 + m3, s34 = m3 + m4, s45 = m4 + m5;
  const u16x2 n3 = (m1 + s23 + s23 + m4 + m4 + m5 + usplat(4)) >> 3;
  const u16x2 n4 = (m2 + s34 + s34 + m5 + m5 + m6 + usplat(4)) >> 3;
  const u16x2 n2 = (m1 + s23 + m4 + usplat(2)) >> 2;
  const u16x2 n5 = (s34 + m5 + m6 + usplat(2)) >> 2;
  const u16x2 n1 = (m0 + m0 + m1 + m1 + m1 + s23 + m4 + usplat(4)) >> 3;
  const u16x2 n6 = (s34 + m5 + m6 + m6 + m6 + m7 + m7 + usplat(4)) >> 3;
  (void)s45;
  auto lim = [&](u16x2 m, u16x2 n) { const s16x2 c = as_s16x2(as_u32(m)); return as_u32(pk_clip(c - t2, c + t2, as_s16x2(as_u32(n)))); };
  s[1] = lim(m1, n1); s[2] = lim(m2, n2); s[3] = lim(m3, n3);
  s[4] = lim(m4, n4); s[5] = lim(m5, n5); s[6] = lim(m6, n6);
}
__device__ inline void filter_luma_pair_weak(uint32_t (&s)[8], int tc, int thr_cut, bool filt_p, bool filt_q, int maxv) {
  const s16x2 m1 = as_s16x2(s[1]), m2 = as_s16x2(s[2]), m3 = as_s16x2(s[3]), m4 = as_s16x2(s[4]), m5 = as_s16x2(s[5]), m6 = as_s16x2(s[6]);
  const s16x2 zero = splat(0), mx = splat(maxv), tcv = splat(tc), tc2 = splat(tc >> 1);
  s16x2 delta = ((m4 - m3) * splat(9) - (m5 - m2) * splat(3) + splat(8)) >> 4;
  // 0xffff in the halves whose line is filtered (|delta| < thr_cut): the sign of |delta| - thr_cut
  const uint32_t on = as_u32((__builtin_elementwise_max(delta, -delta) - splat(thr_cut)) >> 15);
  delta = pk_clip(-tcv, tcv, delta);
  const uint32_t n3 = as_u32(pk_clip(zero, mx, m3 + delta)), n4 = as_u32(pk_clip(zero, mx, m4 - delta));
  s[3] = (n3 & on) | (s[3] & ~on);
  s[4] = (n4 & on) | (s[4] & ~on);
  if (filt_p) {
    const s16x2 d1 = pk_clip(-tc2, tc2, ((((m1 + m3 + splat(1)) >> 1) - m2 + delta) >> 1));
    s[2] = (as_u32(pk_clip(zero, mx, m2 + d1)) & on) | (s[2] & ~on);
  }
  if (filt_q) {
    const s16x2 d2 = pk_clip(-tc2, tc2, ((((m6 + m4 + splat(1)) >> 1) - m5 - delta) >> 1));
    s[5] = (as_u32(pk_clip(zero, mx, m5 + d2)) & on) | (s[5] & ~on);
  }
}

// luma decisions and filtering of one 4-line unit (xEdgeFilterLuma :587-650); a[k] = sample k across the edge of lines 0
// (low half) and 1 (high half), b[k] = the same of lines 2 and 3
__device__ inline void filter_luma_unit(uint32_t (&a)[8], uint32_t (&b)[8], int bs, int qp, int tc_offset_div2, int beta_offset_div2, int bd,
                                        bool p_nofilt = false, bool q_nofilt = false) {
  const int scale = 1 << (bd - 8);
  const int tc = c_tc_table[clip3(0, 53, qp + 2 * (bs - 1) + (tc_offset_div2 << 1))] * scale;
  const int beta = c_beta_table[clip3(0, 51, qp + (beta_offset_div2 << 1))] * scale;
  const int side = (beta + (beta >> 1)) >> 3;
  int l0[8], l3[8];                                   // the decisions look at lines 0 and 3
#pragma unroll
  for (int k = 0; k < 8; k++) { l0[k] = (int)(a[k] & 0xffffu); l3[k] = (int)(b[k] >> 16); }
  const int dp0 = abs(l0[1] - 2 * l0[2] + l0[3]), dq0 = abs(l0[4] - 2 * l0[5] + l0[6]);
  const int dp3 = abs(l3[1] - 2 * l3[2] + l3[3]), dq3 = abs(l3[4] - 2 * l3[5] + l3[6]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
  if (d >= beta) return;
  const bool fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
  const bool s0 = (abs(l0[0] - l0[3]) + abs(l0[7] - l0[4]) < (beta >> 3)) && (2 * d0 < (beta >> 2)) &&
                  (abs(l0[3] - l0[4]) < ((tc * 5 + 1) >> 1));
  const bool s3 = (abs(l3[0] - l3[3]) + abs(l3[7] - l3[4]) < (beta >> 3)) && (2 * d3 < (beta >> 2)) &&
                  (abs(l3[3] - l3[4]) < ((tc * 5 + 1) >> 1));
  const int maxv = (1 << bd) - 1;
  // bPartPNoFilter / bPartQNoFilter (xPelFilterLuma :847-858): a lossless / PCM side keeps its samples
  uint32_t ka[8], kb[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { ka[k] = a[k]; kb[k] = b[k]; }
  if (s0 && s3) {
    filter_luma_pair_strong(a, tc);
    filter_luma_pair_strong(b, tc);
  } else {
    filter_luma_pair_weak(a, tc, tc * 10, fp, fq, maxv);
    filter_luma_pair_weak(b, tc, tc * 10, fp, fq, maxv);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (p_nofilt) { a[k] = ka[k]; b[k] = kb[k]; }
    if (q_nofilt) { a[4 + k] = ka[4 + k]; b[4 + k] = kb[4 + k]; }
  }
}

// lines 0..3 of a vertical-edge unit arrive as rows of eight contiguous samples (four dwords each); transpose them into the
// line-pair form above and back.  v_perm_b32 selectors: 0x05040100 = low halves of (second, first), 0x07060302 = high halves
__device__ inline void rows_to_pairs(const uint32_t (&r0)[4], const uint32_t (&r1)[4], uint32_t (&a)[8]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    a[2 * j] = __builtin_amdgcn_perm(r1[j], r0[j], 0x05040100u);
    a[2 * j + 1] = __builtin_amdgcn_perm(r1[j], r0[j], 0x07060302u);
  }
}
__device__ inline void pairs_to_rows(const uint32_t (&a)[8], uint32_t (&r0)[4], uint32_t (&r1)[4]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    r0[j] = __builtin_amdgcn_perm(a[2 * j + 1], a[2 * j], 0x05040100u);
    r1[j] = __builtin_amdgcn_perm(a[2 * j + 1], a[2 * j], 0x07060302u);
  }
}

// chroma tc for one component (xEdgeFilterChroma :759-775); fmt: chroma_format_idc (the QP table of 4:2:0, min(qPi, 51) otherwise)
__device__ inline int chroma_tc(int qp_avg, int pps_off, int tc_offset_div2, int bd, int fmt = 1) {
  int qp = qp_avg + pps_off;
  if (qp >= 58) qp = fmt == 1 ? qp - 6 : min(qp, 51);
  else if (qp >= 0) qp = fmt == 1 ? c_chroma_scale_420_dbk[qp] : min(qp, 51);
  return c_tc_table[clip3(0, 53, qp + 2 + (tc_offset_div2 << 1))] * (1 << (bd - 8));     // Bs == 2: + DEFAULT_INTRA_TC_OFFSET
}



// neighbour samples x+DX .. x+7+DX of a row as four packed pairs; `e` = the row's 8 samples, l / r = samples x-1 / x+8
template <int DX>
__device__ inline void shifted(const u32x4 e, uint32_t l, uint32_t r, uint32_t (&n)[4]) {
  if constexpr (DX == 0) { n[0] = e.x; n[1] = e.y; n[2] = e.z; n[3] = e.w; }
  else if constexpr (DX < 0) {
    n[0] = (e.x << 16) | l; n[1] = __builtin_amdgcn_alignbit(e.y, e.x, 16);
    n[2] = __builtin_amdgcn_alignbit(e.z, e.y, 16); n[3] = __builtin_amdgcn_alignbit(e.w, e.z, 16);
  } else {
    n[0] = __builtin_amdgcn_alignbit(e.y, e.x, 16); n[1] = __builtin_amdgcn_alignbit(e.z, e.y, 16);
    n[2] = __builtin_amdgcn_alignbit(e.w, e.z, 16); n[3] = (e.w >> 16) | (r << 16);
  }
}

// offsets by table index (two indices 0..7 packed as 16-bit halves) -> two sign-extended 16-bit offsets.  v_perm_b32 does
// the 8-entry byte-table lookup for both halves at once: indices are moved to the odd bytes so that the second v_perm can
// replicate the sign bits (selector codes 8 / 9 = sign of byte 1 / 3).
__device__ inline s16x2 lut_offsets(uint32_t idx_pk, uint32_t tab_lo, uint32_t tab_hi) {
  const uint32_t looked = __builtin_amdgcn_perm(tab_hi, tab_lo, idx_pk << 8);   // bytes 1,3 = table[idx]; bytes 0,2 = table[0] (unused)
  return as_s16x2(__builtin_amdgcn_perm(0u, looked, 0x09030801u));
}

// PCMLFDisableProcess (TComSampleAdaptiveOffset.cpp:742-835) folded into SAO: which of the 8 samples at (x, row) of component
// comp belong to lossless / PCM-unfiltered CUs and keep the SAO input.  Returns a mask with 0xffff per exempt sample pair half
// packed like the samples (4 dwords); only called for pictures that hold such CUs (PicDev::any_nofilt).
__device__ inline void sao_exempt_mask(const PicDev& P, int comp, int x, int row, uint32_t (&m)[4]) {
  const int sx = comp ? P.csx : 0, sy = comp ? P.csy : 0;
  const BlkInfo* g = P.blk + (size_t)((row << sy) >> 2) * P.grid_w + ((x << sx) >> 2);
  // full horizontal resolution: samples 0-3 / 4-7 lie in two 4x4 blocks; subsampled chroma: every pair of samples in its own block (four blocks)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int blk = sx ? j : (j >> 1);
    m[j] = (ldg(&g[blk].flags) & BF_NOFILT) ? 0xffffffffu : 0u;
  }
}

// Chroma planes hold Cb and Cr alternately (hmgpu_dev.h "chroma planes").  Eight consecutive samples of component comp (1 / 2) from
// position x (a multiple of 4) of a row -- `row_c` points at the component's sample of position 0 --, as four packed pairs like a luma load:
__device__ inline u32x4 ldc8(const int16_t* row_c, int x, int comp) {
  const int16_t* p = row_c - (comp - 1) + kCStep * x;
  const u32x4 a = ldg4(p), b = ldg4(p + 8);
  const uint32_t sel = comp == 2 ? 0x07060302u : 0x05040100u;
  return u32x4{__builtin_amdgcn_perm(a.y, a.x, sel), __builtin_amdgcn_perm(a.w, a.z, sel), __builtin_amdgcn_perm(b.y, b.x, sel), __builtin_amdgcn_perm(b.w, b.z, sel)};
}
// ... and back, sample by sample (the other component's halves belong to another thread)
__device__ inline void stc8(int16_t* row_c, int x, const u32x4 v) {
  int16_t* p = row_c + kCStep * x;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; j++) { stg(p + kCStep * 2 * j, (int16_t)(w[j] & 0xffffu)); stg(p + kCStep * (2 * j + 1), (int16_t)(w[j] >> 16)); }
}

// edge-offset arithmetic of one row of 8 samples once the two neighbour rows are at hand (na / nb = the samples at
// (x+DX, row+DY) / (x-DX, row-DY) as four packed pairs): shared by k_sao (neighbours from global memory) and the fused
// filter kernel (neighbours from its LDS tile)
template <int DX, int DY>
__device__ inline void sao_eo_core(int x, int row, const u32x4 cur, const uint32_t (&na)[4], const uint32_t (&nb)[4], uint32_t off_lo, uint32_t off_hi,
                                   unsigned av, int x0, int y0, int x1, int y1, int maxv, uint32_t (&out)[4]) {
  const int ya = row + DY, yb = row - DY;
  // availability: interior samples face positions in the CTB's own columns; sample 0 / the last sample may face the
  // left / right CTU column
  const int va = ya < y0 ? 0 : (ya > y1 ? 2 : 1), vb = yb < y0 ? 0 : (yb > y1 ? 2 : 1);
  // rows of the availability grid (SaoDev::avail) that hold the two compared positions: bit h = column class
  const unsigned ra = av >> (3 * va), rb = av >> (3 * vb);
  const bool mid_ok = ((ra & rb) >> 1) & 1;
  const int last = min(7, x1 - x);                          // last sample of the group that lies inside the CTB / picture
  const int ha0 = (x + DX) < x0 ? 0 : 1, hb0 = (x - DX) < x0 ? 0 : 1;
  const int hal = (x + last + DX) > x1 ? 2 : 1, hbl = (x + last - DX) > x1 ? 2 : 1;
  const bool ok0 = ((ra >> ha0) & (rb >> hb0)) & 1;
  const bool okl = ((ra >> hal) & (rb >> hbl)) & 1;
  const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
  // per-half enable masks.  Groups start at multiples of 8 and component widths are multiples of 4 (luma: 8), so `last` is 7,
  // or 3 for the last chroma group of a picture whose chroma width is 4 mod 8: sample 0 takes ok0, sample `last` okl, the rest
  // (and whatever lies beyond the picture, inside the margin) mid_ok
  const uint32_t midm = mid_ok ? 0xffffffffu : 0u;
  const uint32_t lastm = (midm & 0x0000ffffu) | (okl ? 0xffff0000u : 0u);
  uint32_t m[4] = {midm, midm, midm, midm};
  if constexpr (DX != 0) {                                  // vertical class: no sample faces another CTU column
    m[0] = (midm & 0xffff0000u) | (ok0 ? 0x0000ffffu : 0u);
    m[1] = last == 3 ? lastm : midm;
    m[3] = last == 3 ? midm : lastm;
  }
  // the two clamp bounds are hidden from the optimiser: with visible constants it turns clamp(c - n, -1, 1) into two 16-bit
  // compares and selects per HALF (90 instructions per group instead of 16 packed min / max)
  uint32_t one_u = 0x00010001u, mone_u = 0xffffffffu;
  asm("" : "+v"(one_u));
  asm("" : "+v"(mone_u));
  const s16x2 one = as_s16x2(one_u), mone = as_s16x2(mone_u);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const s16x2 cc = as_s16x2(c[j]);
    const s16x2 sa = __builtin_elementwise_max(__builtin_elementwise_min(cc - as_s16x2(na[j]), one), mone);
    const s16x2 sb = __builtin_elementwise_max(__builtin_elementwise_min(cc - as_s16x2(nb[j]), one), mone);
    const uint32_t et = as_u32(sa + sb + splat(2));         // edge class 0..4 in each half
    const s16x2 off = lut_offsets(et, off_lo, off_hi);
    const s16x2 res = __builtin_elementwise_min(__builtin_elementwise_max(cc + off, splat(0)), splat(maxv));
    out[j] = (as_u32(res) & m[j]) | (c[j] & ~m[j]);
  }
}

}  // namespace hmgpu
