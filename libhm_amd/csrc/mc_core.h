// mc_core.h -- shared pieces of the motion-compensation kernels (k_mc.hip: the LDS-staged picture kernels; k_mc_cells.hip:
// the per-cell register path for 8x4 / 4x8 / 16x4-style PUs and the kernel-level test seam).
#pragma once
#include "hmgpu_dev.h"
#include "itx_core.h"
#include <algorithm>
#include <cstdlib>

namespace hmgpu {

typedef short short2v __attribute__((ext_vector_type(2)));

static __constant__ int8_t c_luma_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0},
                                         {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static __constant__ int8_t c_chroma_taps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                           {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

__device__ inline uint32_t pack_taps(int a, int b) { return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16); }
__device__ inline int dot2(uint32_t samples, uint32_t taps, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, samples), __builtin_bit_cast(short2v, taps), acc, false);
}

// TComDataCU::clipMv (TComDataCU.cpp:3102-3114): clamp against the CU origin
__device__ inline void clip_mv(const PicDev& P, int cu_x, int cu_y, int& mvx, int& mvy) {
  const int ctu = 1 << P.log2ctu;
  mvx = min((P.width + 8 - cu_x - 1) << 2, max((-ctu - 8 - cu_x + 1) * 4, mvx));
  mvy = min((P.height + 8 - cu_y - 1) << 2, max((-ctu - 8 - cu_y + 1) * 4, mvy));
}

// ---- 14-bit intermediate prediction of a W x H tile (W, H even): out[y][x] = HM's "bi" output of xPredInterBlk.
// TAPS = 8 (luma, 2 fraction bits) or 4 (chroma 4:2:0, 3 fraction bits).  Rows stream through registers: a row is
// fetched as dwords (two samples each), re-paired for even and odd output columns with one funnel shift per dword,
// filtered horizontally with TAPS/2 dot2 per output, and every pair of consecutive intermediate rows is folded into the
// vertical accumulators it contributes to (again TAPS/2 dot2 per output).
// STEP 2 (chroma): `ref` is the plane in which Cb and Cr alternate (hmgpu_dev.h "chroma planes"), a dword of a row = one position's
// (Cb, Cr), `half` picks the component; the pairs of neighbouring samples the dot products want come out of two dwords with one byte
// permute, whatever the parity of the window start.
template <int TAPS, int W, int H, int STEP = 1>
__device__ inline void predict14(const int16_t* __restrict__ ref, int pitch, int x0, int y0, int mvx, int mvy, int bd, int (&out)[H][W], int half = 0) {
  constexpr int FB = TAPS == 8 ? 2 : 3;
  constexpr int BEFORE = TAPS / 2 - 1;
  constexpr int ROWS = H + TAPS - 1, COLS = W + TAPS - 1;
  constexpr int ND = (COLS + 2) / 2;                       // dwords per row: COLS samples from an even address, +1 if the start is odd
  constexpr int HT = TAPS / 2;
  const int xf = mvx & ((1 << FB) - 1), yf = mvy & ((1 << FB) - 1);
  const int xs = x0 + (mvx >> FB) - BEFORE, ys = y0 + (mvy >> FB) - BEFORE;
  uint32_t cx[HT], cy[HT];
#pragma unroll
  for (int k = 0; k < HT; k++) {
    cx[k] = TAPS == 8 ? pack_taps(c_luma_taps[xf][2 * k], c_luma_taps[xf][2 * k + 1]) : pack_taps(c_chroma_taps[xf][2 * k], c_chroma_taps[xf][2 * k + 1]);
    cy[k] = TAPS == 8 ? pack_taps(c_luma_taps[yf][2 * k], c_luma_taps[yf][2 * k + 1]) : pack_taps(c_chroma_taps[yf][2 * k], c_chroma_taps[yf][2 * k + 1]);
  }
  const int head = bd >= 12 ? 2 : 14 - bd;                 // max(2, IF_INTERNAL_PREC - bitDepth)
  const int sh1 = 6 - head;
  const int off1 = -(8192 << sh1);
#pragma unroll
  for (int y = 0; y < H; y++)
#pragma unroll
    for (int x = 0; x < W; x++) out[y][x] = 0;
  const int sh_odd = STEP == 1 ? (xs & 1) * 16 : 0;
  const int16_t* base = ref + (ptrdiff_t)ys * pitch + (STEP == 1 ? (xs & ~1) : STEP * xs);
  constexpr int NL = STEP == 1 ? ND : COLS;                // dwords loaded per row
  const uint32_t sel = half ? 0x07060302u : 0x05040100u;
  // every row of the window is requested before any arithmetic starts: one memory latency per tile instead of one per
  // row (left to itself the compiler issues each row's loads ~100 instructions before their use and then waits for them).
  // Loads return in order, so the counted waits let row r be filtered while rows r+1.. are still in flight.
  uint32_t raw[ROWS][NL];
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(base + (ptrdiff_t)r * pitch);
#pragma unroll
    for (int i = 0; i < NL; i++) raw[r][i] = ldg(q + i);
  }
  __builtin_amdgcn_sched_barrier(0);
  int prev[W];
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    uint32_t a[ND + 1], bq[ND];
    if constexpr (STEP == 1) {
#pragma unroll
      for (int i = 0; i < ND; i++) a[i] = raw[r][i];
      a[ND] = 0;
      // a[j] = (s[2j], s[2j+1]) relative to xs: drop one sample when the start is odd
#pragma unroll
      for (int i = 0; i < ND; i++) a[i] = __builtin_amdgcn_alignbit(a[i + 1], a[i], sh_odd);
      // bq[j] = (s[2j+1], s[2j+2])
#pragma unroll
      for (int i = 0; i + 1 < ND; i++) bq[i] = __builtin_amdgcn_alignbit(a[i + 1], a[i], 16);
      bq[ND - 1] = 0;
    } else {
      // raw[i] = (Cb, Cr) of position xs + i
      uint32_t pr[2 * ND + 1];
#pragma unroll
      for (int i = 0; i < 2 * ND + 1; i++) pr[i] = i < NL ? raw[r][i < NL ? i : 0] : 0u;
#pragma unroll
      for (int i = 0; i < ND; i++) {
        a[i] = __builtin_amdgcn_perm(pr[2 * i + 1], pr[2 * i], sel);
        bq[i] = __builtin_amdgcn_perm(pr[2 * i + 2], pr[2 * i + 1], sel);
      }
      a[ND] = 0;
    }
    int t[W];
#pragma unroll
    for (int x = 0; x < W; x++) {
      int sum = off1;
#pragma unroll
      for (int k = 0; k < HT; k++) sum = dot2((x & 1) ? bq[x / 2 + k] : a[x / 2 + k], cx[k], sum);
      t[x] = sum >> sh1;                                   // HM: filter<N,false,true,false>, a 16-bit Pel
    }
    if (r > 0) {
      // pair (row r-1, row r) feeds output rows y with r-1-y in {0, 2, .., TAPS-2}
#pragma unroll
      for (int x = 0; x < W; x++) {
        const uint32_t pv = __builtin_amdgcn_perm((uint32_t)t[x], (uint32_t)prev[x], 0x05040100u);   // (prev.lo16, t.lo16)
#pragma unroll
        for (int k = 0; k < HT; k++) {
          const int y = r - 1 - 2 * k;
          if (y >= 0 && y < H) out[y][x] = dot2(pv, cy[k], out[y][x]);
        }
      }
    }
#pragma unroll
    for (int x = 0; x < W; x++) prev[x] = t[x];
  }
  // out[][] still carries the 6 fractional bits of the vertical pass: HM's value is out >> 6 (filter<N,true,false,false>)
}

// uni-prediction final rounding, v6 = vertical sum with its 6 fractional bits: HM computes ((v6 >> 6) + 8192 + 2^(head-1)) >> head
// [filter isLast]; the constants are multiples of 64, so both shifts merge exactly into one
__device__ inline int finish_uni(int v6, int head, int maxv) { return clip3(0, maxv, (v6 + (8192 << 6) + (32 << head)) >> (6 + head)); }
// TComYuv::addAvg: clip((a + b + 2^head + 2*8192) >> (head+1))
__device__ inline int finish_bi(int a, int b, int head, int maxv) { return clip3(0, maxv, (a + b + (1 << head) + 16384) >> (head + 1)); }

// explicit weighted prediction of one lane's block (TComWeightPrediction.cpp:44-57, 211-271): weight / offset per list for this
// component, log2 of the weight denominator; active = the block's slice uses it (TComSlice::applyWP)
struct WpLane { bool active; int w[2], o[2], log2wd; };

// the reference index of a 4x4 block: BlkInfo keeps the reference PICTURE (what the filter needs), weighted prediction is
// indexed by reference INDEX (two indices may name the same picture with different weights): read it back from HM's array
__device__ inline int ref_idx_at(const PicDev& P, int list, int x, int y) {
  const int m = (1 << P.log2ctu) - 1;
  const int ctu = (y >> P.log2ctu) * P.ctus_w + (x >> P.log2ctu);
  const int bx = (x & m) >> 2, by = (y & m) >> 2;
  int z = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) z |= (((bx >> k) & 1) << (2 * k)) | (((by >> k) & 1) << (2 * k + 1));
  return ldg(P.ref_idx[list] + (size_t)ctu * P.parts + z);
}
__device__ inline WpLane wp_lane(const PicDev& P, const BlkInfo& bi, int comp, int x, int y) {
  WpLane w;
  const SliceDev& sd = P.slices[bi.slice];
  w.active = ldg(&sd.weighted_pred) != 0;
  w.w[0] = w.w[1] = 1; w.o[0] = w.o[1] = 0; w.log2wd = 0;
  if (w.active) {
    w.log2wd = ldg(&sd.wp_log2_denom[comp ? 1 : 0]);
#pragma unroll
    for (int l = 0; l < 2; l++)
      if (bi.flags & (l ? BF_MC_L1 : BF_MC_L0)) {
        const int r = ref_idx_at(P, l, x, y);
        w.w[l] = ldg(&sd.wp_weight[l][r][comp]);
        w.o[l] = ldg(&sd.wp_offset[l][r][comp]);
      }
  }
  return w;
}

// prediction of one W x H tile of component `comp` with the motion of `bi`, written to dst (the picture being decoded) with the
// residual added when the tile carries one (coded; PicDev::resid, written by k_itx; W x H lies inside one 4x4 block)
// mvsx / mvsy (chroma of 4:2:2 / 4:4:4 pictures): the vector is in quarter LUMA samples = 1 / (4 << cs) of the component's; the 4-tap
// filters are indexed in eighths: shifted up by 1 - cs (TComPrediction.cpp:664-674, TComInterpolationFilter.cpp:344-346)
template <int TAPS, int W, int H, bool WP = false>
__device__ inline void predict_tile(const PicDev& P, const PlaneSet* __restrict__ finals, int comp, int x0, int y0,
                                    const BlkInfo& bi, int cu_x, int cu_y, int16_t* __restrict__ dst, int lx = 0, int ly = 0, bool coded = false,
                                    int mvsx = 0, int mvsy = 0) {
  WpLane wp = {false, {1, 1}, {0, 0}, 0};
  if constexpr (WP) wp = wp_lane(P, bi, comp, lx, ly);
  const int bd = P.bd[comp];
  const int head = bd >= 12 ? 2 : 14 - bd;
  const int maxv = (1 << bd) - 1;
  const int pitch = P.pitch[comp];
  uint32_t res[H][W / 2];                                  // finished samples, two per register
  const int l0 = (bi.flags & BF_MC_L0) ? 0 : 1;
  const bool both = (bi.flags & (BF_MC_L0 | BF_MC_L1)) == (BF_MC_L0 | BF_MC_L1);
  {
    int a[H][W];
    // (selected, not indexed: a run-time index into the record would put it into scratch memory -- 64 bytes per lane written out
    // for every tile, measured as +50 % / +100 % on the luma / chroma kernels' write traffic)
    int mvx = l0 ? bi.mv[1][0] : bi.mv[0][0], mvy = l0 ? bi.mv[1][1] : bi.mv[0][1];
    const int ref0 = l0 ? bi.ref[1] : bi.ref[0];
    clip_mv(P, cu_x, cu_y, mvx, mvy);
    mvx <<= mvsx; mvy <<= mvsy;
    if (comp == 0) predict14<TAPS, W, H>(ldg(&finals[ref0].p[0]), pitch, x0, y0, mvx, mvy, bd, a);
    else predict14<TAPS, W, H, kCStep>(ldg(&finals[ref0].p[1]), pitch, x0, y0, mvx, mvy, bd, a, comp - 1);
#pragma unroll
    for (int y = 0; y < H; y++)
#pragma unroll
      for (int x = 0; x < W; x += 2) {
        // bi: park the 14-bit intermediates (they fit 16 bits) while the second list is computed
        int v0 = both ? (a[y][x] >> 6) : finish_uni(a[y][x], head, maxv), v1 = both ? (a[y][x + 1] >> 6) : finish_uni(a[y][x + 1], head, maxv);
        if (WP && wp.active && !both) {
          // weightUnidir on HM's 14-bit intermediate (xPredInterUni with bi = true, then addWeightUni)
          const int shift = wp.log2wd + head, round = shift > 0 ? 1 << (shift - 1) : 0;
          const int ww = l0 ? wp.w[1] : wp.w[0], wo = l0 ? wp.o[1] : wp.o[0];
          v0 = clip3(0, maxv, ((ww * ((a[y][x] >> 6) + 8192) + round) >> shift) + wo);
          v1 = clip3(0, maxv, ((ww * ((a[y][x + 1] >> 6) + 8192) + round) >> shift) + wo);
        }
        res[y][x / 2] = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x05040100u);
      }
  }
  if (both) {
    int b[H][W];
    int mvx = bi.mv[1][0], mvy = bi.mv[1][1];
    clip_mv(P, cu_x, cu_y, mvx, mvy);
    mvx <<= mvsx; mvy <<= mvsy;
    if (comp == 0) predict14<TAPS, W, H>(ldg(&finals[bi.ref[1]].p[0]), pitch, x0, y0, mvx, mvy, bd, b);
    else predict14<TAPS, W, H, kCStep>(ldg(&finals[bi.ref[1]].p[1]), pitch, x0, y0, mvx, mvy, bd, b, comp - 1);
#pragma unroll
    for (int y = 0; y < H; y++)
#pragma unroll
      for (int x = 0; x < W; x += 2) {
        const int a0 = (int)(int16_t)(res[y][x / 2] & 0xffffu), a1 = (int)(int16_t)(res[y][x / 2] >> 16);
        int v0 = finish_bi(a0, b[y][x] >> 6, head, maxv), v1 = finish_bi(a1, b[y][x + 1] >> 6, head, maxv);
        if (WP && wp.active) {
          // weightBidir (addWeightBi): shift = log2Wd + 1 + shiftNum, the offsets of both lists enter at half weight
          const int shift = wp.log2wd + 1 + head, add = (1 << (shift - 1)) + ((wp.o[0] + wp.o[1]) << (shift - 1));
          v0 = clip3(0, maxv, (wp.w[0] * (a0 + 8192) + wp.w[1] * ((b[y][x] >> 6) + 8192) + add) >> shift);
          v1 = clip3(0, maxv, (wp.w[0] * (a1 + 8192) + wp.w[1] * ((b[y][x + 1] >> 6) + 8192) + add) >> shift);
        }
        res[y][x / 2] = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x05040100u);
      }
  }
  if (coded) {
    const int rtw = (P.grid_w * 4 >> (comp ? P.csx : 0)) >> 3;
    const uint32_t maxv2 = (uint32_t)maxv * 0x10001u;
#pragma unroll
    for (int y = 0; y < H; y++) {
      const int yy = y0 + y;
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(P.resid[comp] + (((size_t)(yy >> 3) * rtw + (x0 >> 3)) * 8 + resid_slot(yy)) * 8 + (x0 & 7));
#pragma unroll
      for (int x = 0; x < W / 2; x++) res[y][x] = pk_clip_u(pk_add_sat(res[y][x], ldg(rp + x)), maxv2);
    }
  }
  if (comp) {
    // a chroma component: every other element of its plane (dst = PicDev::rec[comp])
#pragma unroll
    for (int y = 0; y < H; y++) {
      int16_t* row = dst + (ptrdiff_t)(y0 + y) * pitch + kCStep * x0;
#pragma unroll
      for (int x = 0; x < W / 2; x++) { stg(row + kCStep * 2 * x, (int16_t)(res[y][x] & 0xffffu)); stg(row + kCStep * (2 * x + 1), (int16_t)(res[y][x] >> 16)); }
    }
    return;
  }
#pragma unroll
  for (int y = 0; y < H; y++) {
    int16_t* row = dst + (ptrdiff_t)(y0 + y) * pitch + x0;
    if constexpr (W == 8) { u32x4 v = {res[y][0], res[y][1], res[y][2], res[y][3]}; stg4(row, v); }
    else if constexpr (W == 4) { u32x2 v = {res[y][0], res[y][1]}; stg2(row, v); }
    else stg(reinterpret_cast<uint32_t*>(row), res[y][0]);
  }
}

__device__ inline bool same_motion(const BlkInfo& a, const BlkInfo& b) {
  const uint4 ua = *reinterpret_cast<const uint4*>(&a), ub = *reinterpret_cast<const uint4*>(&b);
  // mv[2][2], ref[2] and the MC flag bits + CU size (the CU origin enters clipMv)
  return ua.x == ub.x && ua.y == ub.y && (ua.z & 0xffff) == (ub.z & 0xffff) &&
         ((a.flags ^ b.flags) & (BF_VALID | BF_INTRA | BF_MC_L0 | BF_MC_L1)) == 0 && a.log2cu == b.log2cu;
}
__device__ inline bool is_inter(const BlkInfo& b) { return (b.flags & BF_VALID) && (b.flags & (BF_MC_L0 | BF_MC_L1)); }

// one 4x4 luma cell / its 2x2 chroma samples on their own: only where the four cells of an 8x8 area do not share
// their motion (8x4 / 4x8 PUs, AMP parts of 16x16 CUs, picture borders).  Out of line: rare, and it keeps the common
// path's register budget small.
// rmask: the TR_* residual mask of the 8x8 tile the cell lies in
template <bool WP>
__device__ __attribute__((noinline)) void luma_cell(const PicDev& P, const PlaneSet* __restrict__ finals, const BlkInfo c, int x, int y, uint32_t rmask) {
  const int cs = 1 << c.log2cu;
  predict_tile<8, 4, 4, WP>(P, finals, 0, x, y, c, x & ~(cs - 1), y & ~(cs - 1), P.rec[0], x, y, ((rmask >> (((y >> 2) & 1) * 2 + ((x >> 2) & 1))) & 1) != 0);
}
template <bool WP>
__device__ __attribute__((noinline)) void chroma_cell(const PicDev& P, const PlaneSet* __restrict__ finals, const BlkInfo c, int lx, int ly, uint32_t rmask) {
  const int cs = 1 << c.log2cu;
  predict_tile<4, 2, 2, WP>(P, finals, 1, lx >> 1, ly >> 1, c, lx & ~(cs - 1), ly & ~(cs - 1), P.rec[1], lx, ly, (rmask & TR_CB) != 0);
  predict_tile<4, 2, 2, WP>(P, finals, 2, lx >> 1, ly >> 1, c, lx & ~(cs - 1), ly & ~(cs - 1), P.rec[2], lx, ly, (rmask & TR_CR) != 0);
}

// the chroma of one 4x4 luma cell of a 4:2:2 / 4:4:4 picture: (4 >> CSX) x (4 >> CSY) samples of both components, the residual added from
// the tiles (zero where no coded block lies: hmgpu_api.hip clears them for these formats)
template <bool WP, int CSX, int CSY>
__device__ __attribute__((noinline)) void chroma_cell_fmt(const PicDev& P, const PlaneSet* __restrict__ finals, const BlkInfo c, int lx, int ly) {
  const int cs = 1 << c.log2cu;
  predict_tile<4, (4 >> CSX), (4 >> CSY), WP>(P, finals, 1, lx >> CSX, ly >> CSY, c, lx & ~(cs - 1), ly & ~(cs - 1), P.rec[1], lx, ly, true, 1 - CSX, 1 - CSY);
  predict_tile<4, (4 >> CSX), (4 >> CSY), WP>(P, finals, 2, lx >> CSX, ly >> CSY, c, lx & ~(cs - 1), ly & ~(cs - 1), P.rec[2], lx, ly, true, 1 - CSX, 1 - CSY);
}

// lane -> 8x8 luma area: a wave covers 8x8 areas = 64x64 luma samples, a block four such squares in CTU order
__device__ inline bool tile_origin(const PicDev& P, const Batch& b, int slot, int lb, int& x0, int& y0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ctu_sz = 1 << P.log2ctu;
  const int sq = max(1, ctu_sz / 64);                      // 64x64 squares per CTU row
  const int per_ctu = sq * sq;
  const int sid = lb * 4 + wave;
  if (sid >= b.num_ctus[slot] * per_ctu) return false;
  const int ctu = b.first_ctu[slot] + sid / per_ctu;
  const int s_in = sid % per_ctu;
  const int cx = (ctu % P.ctus_w) * ctu_sz, cy = (ctu / P.ctus_w) * ctu_sz;
  x0 = cx + (s_in % sq) * 64 + (lane & 7) * 8;
  y0 = cy + (s_in / sq) * 64 + (lane >> 3) * 8;
  return x0 < cx + ctu_sz && y0 < cy + ctu_sz && x0 < P.width && y0 < P.height;
}

// with explicit weighted prediction two cells share a tile only if their reference INDICES agree too
template <bool WP>
__device__ inline bool tile_is_uniform(const PicDev& P, const BlkInfo& c00, const BlkInfo& c01, const BlkInfo& c10, const BlkInfo& c11, int x0, int y0) {
  if (!(is_inter(c00) && same_motion(c00, c01) && same_motion(c00, c10) && same_motion(c00, c11))) return false;
  if constexpr (WP) {
    if (ldg(&P.slices[c00.slice].weighted_pred)) {
#pragma unroll
      for (int l = 0; l < 2; l++)
        if (c00.flags & (l ? BF_MC_L1 : BF_MC_L0)) {
          const int r = ref_idx_at(P, l, x0, y0);
          if (ref_idx_at(P, l, x0 + 4, y0) != r || ref_idx_at(P, l, x0, y0 + 4) != r || ref_idx_at(P, l, x0 + 4, y0 + 4) != r) return false;
        }
    }
  }
  return true;
}

}  // namespace hmgpu
