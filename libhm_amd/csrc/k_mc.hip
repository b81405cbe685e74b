// k_mc.hip -- motion compensation (TComPrediction::motionCompensation -> xPredInterBlk -> TComInterpolationFilter,
// TComYuv::addAvg; TComPrediction.cpp:514-714, TComInterpolationFilter.cpp:166-251, TComYuv.cpp:336-391).
//
// One 256-thread workgroup predicts one 64x64 luma square (a CTU, or the part of the picture a smaller CTU covers) of one
// picture of the batch.  The unit of bookkeeping is the 8x8 luma TILE (the 4x4 chroma tile of both planes under it), whose
// clipped motion k_prep has written as one 16-byte TileMv record; the unit of sharing is the vertical RUN of tiles with
// identical motion -- whatever PU, CU or merge candidate that motion came from: prediction depends on position and motion
// only, and the window rows of a tile that its upper neighbour of the same run has filtered horizontally are not
// filtered again.
//
//   prologue every wave, lane = tile: the 64 TileMv records of the square and the context's table of final planes (lane =
//            device picture) in ONE round of loads; "top of its run" by comparing with the record eight lanes up; the run
//            tops compacted with one ds_permute.  Everything a thread needs about its work items is then a lane shuffle
//            away: no LDS, no barrier, no dependent global load before the window loads.
//   H pass   HM's filter<N, isVertical = false, isFirst, !isLast> (TComInterpolationFilter.cpp:166-251).  A work item is
//            two consecutive window rows x 8 columns (luma) / x 4 columns of both planes (chroma) of one tile, read
//            straight from the reference picture with dword-aligned 16-byte loads: the eight tiles of a tile row sit in
//            adjacent lanes, so lanes whose tiles share their motion read one contiguous row segment and every reference
//            row of a run is fetched once (the register path this replaces fetched 15 window rows per 8 output rows).
//            Every tile owns the 8 (4) window rows below its first 8 (4); a run's top tile also owns those first rows
//            ("halo").  The two 16-bit results of a column go to LDS as ONE dword (row 2i, row 2i+1): the layout the
//            vertical pass multiplies with v_dot2_i32_i16, so the transposition costs nothing.
//   V pass   filter<N, isVertical = true, !isFirst, isLast> / the 14-bit output of bi-prediction + TComYuv::addAvg.  A
//            thread produces two output rows x 8 (4) columns from five (three) row pairs in LDS: the even row with the
//            taps paired (c0,c1)(c2,c3).., the odd row with (0,c0)(c1,c2)..(c7,0) over the same pairs.
//   The same pairing trick serves the H pass: windows start at the even sample at or before their first sample (dword
//   loads), and whether an output column starts in the low or the high half of a dword only selects between the two tap
//   pairings, per lane, from a table -- no funnel shifts, no phase- or parity-dependent branches.
//   bi-prediction (BI variants, launched for batches that hold B slices) runs the two passes once per list; the first
//   list's 14-bit result waits in registers.
//
// Arithmetic: always H then V with the phase-0 filter {0,0,0,64,0,0,0,0} standing for "no interpolation in this
// direction": exact, because HM's one-pass cases are the two-pass formula with one pass a multiplication by 64 whose
// shift commutes with the floor (DESIGN.md "MC arithmetic").  HM's -8192 offset of the 14-bit intermediate (IF_INTERNAL_OFFS)
// is not carried through LDS: the taps sum to 64, so it re-enters as one constant where HM's formulas need it.
// Tiles whose four cells do not share their motion (8x4 / 4x8 PUs, AMP parts of 16x16 CUs) are left to k_mc_cells.hip.
#include "mc_core.h"
#include "itx_core.h"
#include <type_traits>

namespace hmgpu {

typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
#if defined(MC_EXP) && (MC_EXP & 1)         // experiment: no window loads
__device__ inline u32x4 ldg4_a4(const void* p) { const uint32_t v = (uint32_t)(uintptr_t)p; return u32x4{v, v + 1, v + 2, v + 3}; }
#elif defined(MC_NT)
__device__ inline u32x4 ldg4_a4(const void* p) { return __builtin_nontemporal_load((const u32x4_a4 HMGPU_AS1*)p); }
#else
__device__ inline u32x4 ldg4_a4(const void* p) { return *(const u32x4_a4 HMGPU_AS1*)p; }
#endif
#ifndef MC_LB_LUMA
#define MC_LB_LUMA 7      // waves per SIMD the uni-prediction kernel is held to (72 registers; 8 would spill the residual rows)
#endif
#ifndef MC_LB_CHROMA
#define MC_LB_CHROMA 8
#endif

__device__ inline uint32_t pk_sub(uint32_t a, uint32_t b) {                           // per half: a - b (wrapping)
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}


// ---- tap tables: entry (fraction f, parity p) of an N-tap filter c[]: with A = (c0,c1)(c2,c3).. and B = (0,c0)(c1,c2)..(c[N-1],0)
//   p = 0: even columns A,0   odd columns B        p = 1: even columns B   odd columns 0,A
// (column x of a window that starts in the high half of its first dword begins one sample later than its dword)
struct alignas(16) TapsLuma { uint32_t e[12][12]; };   // fractions 0..3, then the identity at tap 0 (4) and at tap 4 (5)
struct alignas(16) TapsChroma { uint32_t e[18][8]; };   // fractions 0..7, then the identity at tap 2 (8)
constexpr uint32_t pk16(int a, int b) { return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16); }
constexpr TapsLuma make_taps_luma() {
  constexpr int c[6][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1},
                           {64, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 64, 0, 0, 0}};
  TapsLuma t = {};
  for (int f = 0; f < 6; f++)
    for (int p = 0; p < 2; p++) {
      uint32_t A[5] = {}, B[5] = {};
      for (int j = 0; j < 4; j++) A[j] = pk16(c[f][2 * j], c[f][2 * j + 1]);
      B[0] = pk16(0, c[f][0]);
      for (int j = 1; j < 4; j++) B[j] = pk16(c[f][2 * j - 1], c[f][2 * j]);
      B[4] = pk16(c[f][7], 0);
      for (int j = 0; j < 5; j++) { t.e[f * 2 + p][j] = p ? B[j] : A[j]; t.e[f * 2 + p][5 + j] = p ? (j == 0 ? 0u : A[j - 1]) : B[j]; }
    }
  return t;
}
constexpr TapsChroma make_taps_chroma() {
  constexpr int c[9][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2},
                           {0, 0, 64, 0}};
  TapsChroma t = {};
  for (int f = 0; f < 9; f++)
    for (int p = 0; p < 2; p++) {
      const uint32_t A[3] = {pk16(c[f][0], c[f][1]), pk16(c[f][2], c[f][3]), 0u};
      const uint32_t B[3] = {pk16(0, c[f][0]), pk16(c[f][1], c[f][2]), pk16(c[f][3], 0)};
      for (int j = 0; j < 3; j++) { t.e[f * 2 + p][j] = p ? B[j] : A[j]; t.e[f * 2 + p][3 + j] = p ? (j == 0 ? 0u : A[j - 1]) : B[j]; }
    }
  return t;
}
__device__ const TapsLuma g_taps_luma = make_taps_luma();
__device__ const TapsChroma g_taps_chroma = make_taps_chroma();

// first link of a dot2 chain without a register to seed it (the compiler only knows the accumulate-in-place form)
__device__ inline int dot2_first(uint32_t samples, uint32_t taps) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(samples), "v"(taps));
  return d;
}

__device__ inline int dot2_seed(uint32_t samples, uint32_t taps, int seed) {       // seed: wave-uniform (a scalar register)
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(samples), "v"(taps), "s"(seed));
  return d;
}
// (lo >> sh) into the low half, (hi >> sh) into the high half of one register: two sub-dword writes instead of two shifts and a
// byte permute.  sh lives in a vector register (SDWA operands).
__device__ inline uint32_t pack_shr(int lo, int hi, int sh) {
  uint32_t d;
  asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(d) : "v"(sh), "v"(lo));
  asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(d) : "v"(sh), "v"(hi));
  return d;
}

// LDS: H-pass results as 8-byte pieces (two columns x one row pair).  Every wave owns a STRIP of the square (two tile rows) and a
// slice of LDS of its own, laid out [tile row][piece][row pair][tile column] so that every DS instruction of both passes touches
// each bank once (the 16 / 32 lanes an LDS cycle serves are the eight tiles of a tile row times two / four row pairs).
//   luma:   piece c = columns 2c, 2c+1 (c < 4), row pair g < 4:            dword (r, c, g, tx) = r * 256 + c * 64 + g * 16 + tx * 2
//   chroma: plane p, piece c = columns 2c, 2c+1 (c < 2), row pair g < 2:   dword (r, p, c, g, tx) = r * 128 + p * 64 + c * 32 + g * 16 + tx * 2
template <bool CHROMA> struct McLds {
  static constexpr int ROW_DW = CHROMA ? 128 : 256;           // dwords per tile row
  struct { uint32_t body[2 * ROW_DW], halo[2 * ROW_DW]; } w[4];   // halo: the first 8 (4) window rows of run tops, at the top tile's position
  uint32_t taps[144];                           // the tap table (every wave writes the same values, reads its own)
  uint32_t htap[CHROMA ? 36 : 1];               // chroma H pass: the halved taps of fraction f as (c/2, c/2) pairs at [4 f], f = 8: the identity at tap 0
#ifdef MC_LDS_PAD
  uint32_t pad[MC_LDS_PAD / 4];                               // experiment: fewer workgroups per CU
#endif
};

struct Square { int sx, sy, ext; };                           // luma origin, extent (min(CTU size, 64)) of the square
__device__ inline bool square_of_block(const McArgs& a, int& slot, Square& g) {
  int lb;
  if (a.mode) { slot = blockIdx.x & (a.n - 1); lb = (int)(blockIdx.x >> a.log2n) * a.per + (int)blockIdx.y; }
  else { slot = blockIdx.x; lb = blockIdx.y; }
  if (lb >= a.num_ctus[slot]) return false;
  const int ctu = a.first_ctu[slot] + lb, ctu_sz = 1 << a.log2ctu;
  g.sx = (ctu % a.ctus_w) * ctu_sz; g.sy = (ctu / a.ctus_w) * ctu_sz; g.ext = min(ctu_sz, 64);
  return true;
}

// weights of one thread's tile (TComWeightPrediction.cpp:44-57, 211-271): slot 0 / slot 1
struct WpTile { bool active; int w0, o0, w1, o1, log2wd; };
template <bool WP>
__device__ inline WpTile wp_tile(const SliceDev* __restrict__ slices, uint32_t flags, uint32_t wpw, int comp) {
  WpTile w = {false, 1, 0, 1, 0, 0};
  if constexpr (WP) {
    const SliceDev& sd = slices[wpw >> 16];
    w.active = ldg(&sd.weighted_pred) != 0;
    if (w.active) {
      w.log2wd = ldg(&sd.wp_log2_denom[comp ? 1 : 0]);
      const int l0 = (flags & TM_FIRST_L1) ? 1 : 0, r0 = wpw & 15, r1 = (wpw >> 4) & 15;
      w.w0 = ldg(&sd.wp_weight[l0][r0][comp]); w.o0 = ldg(&sd.wp_offset[l0][r0][comp]);
      if (flags & TM_BI) { w.w1 = ldg(&sd.wp_weight[1][r1][comp]); w.o1 = ldg(&sd.wp_offset[1][r1][comp]); }
    }
  }
  return w;
}

// the end of a V-pass thread: v6[r][x] = vertical sums of two output rows WITHOUT HM's intermediate offset: v6 = HM's sum + 8192 * 64.
//   pass 0, uni: final samples (filter isLast / weightUnidir) -> dst;  pass 0, bi: HM's 14-bit values (16-bit Pel) wait in park[];
//   pass 1: addAvg / weightBidir with the parked first list -> dst
// rsd: the residual of the two rows (zero where the tile carries none): reconstruction = ClipBD(prediction + residual), TComYuv::addClip
// IL (chroma): lanes l and l + 32 hold the same rows of Cb and Cr; dst = where THIS lane's row (row 0 below 32, row 1 above) starts in the
// plane in which the two components alternate
template <int W, bool WP, bool BI, bool IL = false>
__device__ inline void finish_rows(int (&v6)[2][W], uint32_t (&park)[W], int pass, bool bi, int bd, const WpTile& wp,
                                   const uint32_t (&rsd)[2][W / 2], int16_t* __restrict__ dst, int pitch) {
  const int head = bd >= 12 ? 2 : 14 - bd;
  const uint32_t maxv2 = (uint32_t)((1 << bd) - 1) * 0x10001u;
  if (BI && pass == 0 && bi) {
#pragma unroll
    for (int x = 0; x < W; x++)
      park[x] = pk_sub(__builtin_amdgcn_perm((uint32_t)(v6[1][x] >> 6), (uint32_t)(v6[0][x] >> 6), 0x05040100u), 0x20002000u);
    return;
  }
  uint32_t res[2][W / 2];
  if constexpr (!WP && !BI) {
    // uni-prediction without weights (the rounding constant 2^(5 + head) seeded the sums): shift, pack, ClipBD -- two samples per operation
    const int shv = 6 + head;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int x = 0; x < W; x += 2) res[r][x / 2] = pk_clip_u(pack_shr(v6[r][x], v6[r][x + 1], shv), maxv2);
  } else
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int x = 0; x < W; x += 2) {
      int o[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int v = v6[r][x + i];
        if (!BI || pass == 0) {
          // filter<N, isVertical, !isFirst, isLast>: (HM's sum + 8192 * 64 + 2^(5 + head)) >> (6 + head)
          o[i] = (v + (32 << head)) >> (6 + head);        // (BI / WP variants: the sums start at 0)
          if (WP && wp.active) {
            // weightUnidir on HM's 14-bit intermediate (xPredInterUni with bi = true, then addWeightUni): v >> 6 = intermediate + 8192
            const int shift = wp.log2wd + head, round = shift > 0 ? 1 << (shift - 1) : 0;
            o[i] = ((wp.w0 * (v >> 6) + round) >> shift) + wp.o0;
          }
        } else {
          const uint32_t pk = park[x + i];
          const int a = r ? (int)pk >> 16 : (int)(int16_t)(pk & 0xffffu);
          // TComYuv::addAvg: (a + b + 2^head + 2 * 8192) >> (head + 1) with b = (v >> 6) - 8192
          o[i] = (a + (v >> 6) + (1 << head) + 8192) >> (head + 1);
          if (WP && wp.active) {
            // weightBidir (addWeightBi): shift = log2Wd + 1 + shiftNum, the offsets of both lists enter at half weight
            const int shift = wp.log2wd + 1 + head, add = (1 << (shift - 1)) + ((wp.o0 + wp.o1) << (shift - 1));
            o[i] = (wp.w0 * (a + 8192) + wp.w1 * (v >> 6) + add) >> shift;
          }
        }
      }
      // ClipBD on the packed pair (the saturating pack only matters for weighted prediction: everything else fits 16 bits)
      res[r][x / 2] = pk_clip_u(cvt_pk_sat(o[0], o[1]), maxv2);
    }
#if defined(MC_EXP) && (MC_EXP & 2)         // experiment: no stores (unless a value nobody produces shows up)
  if (res[0][0] != 0x7fff7fffu) return;
#endif
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int x = 0; x < W / 2; x++) res[r][x] = pk_clip_u(pk_add_sat(res[r][x], rsd[r][x]), maxv2);
  if constexpr (IL) {
    static_assert(W == 4, "chroma tiles");
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      // v_permlane32_swap: [0] = Cb, [1] = Cr of this lane's row (below 32: own row 0 | the upper lane's row 0; above: the lower lane's row 1 | own row 1)
      const auto sw = __builtin_amdgcn_permlane32_swap(res[0][j], res[1][j], false, false);
      o[2 * j] = __builtin_amdgcn_perm(sw[1], sw[0], 0x05040100u);
      o[2 * j + 1] = __builtin_amdgcn_perm(sw[1], sw[0], 0x07060302u);
    }
    stg4(dst, u32x4{o[0], o[1], o[2], o[3]});
    return;
  }
#pragma unroll
  for (int r = 0; r < 2; r++) {
    int16_t* row = dst + (ptrdiff_t)r * pitch;
    if constexpr (W == 8) { u32x4 v = {res[r][0], res[r][1], res[r][2], res[r][3]}; stg4(row, v); }
    else { u32x2 v = {res[r][0], res[r][1]}; stg2(row, v); }
  }
}

// the TileMv record of the tile (tx, strip row r) a lane works on, and whether that tile starts a vertical run inside the strip.
// Lanes l and l ^ 32 (luma) / l ^ 16 (chroma) hold the two tiles of a column: one bpermute per dword tells a second-row tile
// whether it continues the first-row tile (same lists, pictures, vectors: the windows are then exactly one tile apart).
__device__ inline u32x4 load_tile_rec(const McArgs& a, int slot, const Square& g, int x, int y) {
  const bool tin = x < g.sx + g.ext && y < g.sy + g.ext && x < a.width && y < a.height;
  u32x4 tm = {0, 0, 0, 0};
  if (tin) tm = ldg4(a.tmv[slot] + (size_t)(y >> 3) * a.tw + (x >> 3));
  return tm;
}
__device__ inline bool tile_is_top(const u32x4 tm, int r, int partner_lane) {
  const uint32_t u0 = __shfl((int)tm.x, partner_lane), u1 = __shfl((int)tm.y, partner_lane), u2 = __shfl((int)tm.z, partner_lane);
  return r == 0 || !(u0 == tm.x && u1 == tm.y && u2 == tm.z);
}

// address of sample (0,0) of the final luma / Cb plane of device picture `ref` (all pictures of a context live in one slab, a
// finished picture is its SAO output or, without SAO, its reconstruction)
// Window loads.  BUF: buffer loads with 32-bit byte offsets from the slab of pictures; a lane without a work item gets an offset past
// the end of the buffer -- the range check answers it with zeros and nothing goes to memory (slabs under kWinLimit bytes: launch_mc).
// !BUF: plain 64-bit addresses; lanes without a work item read valid memory nobody looks at.
constexpr uint32_t kWinLimit = 0xffff0000u;
constexpr uint32_t kResidLimit = 0x7fffff00u;   // residual tiles of a picture: far below 2 GB
template <bool BUF> struct Win;
template <> struct Win<true> {
  typedef uint32_t ref;
  __amdgpu_buffer_rsrc_t rs;
  uint32_t stride;
  __device__ inline Win(const McArgs& a, int) : rs(__builtin_amdgcn_make_buffer_rsrc((void*)a.slab, 0, kWinLimit, 0x00020000)), stride((uint32_t)a.pic_stride) {}
  __device__ inline ref nowhere() const { return kWinLimit; }
  __device__ inline ref plane(const McArgs& a, int pic) const {
    const uint32_t m = pic & 32 ? a.sao_mask_hi : a.sao_mask_lo;
    return (uint32_t)pic * stride + (__builtin_amdgcn_ubfe(m, (uint32_t)pic & 31u, 1u) ? a.sao_off : 0u) + a.origin_off;
  }
  __device__ inline ref at(ref pl, int row, int pitch, int x) const { return pl + (uint32_t)((row * pitch + x) * 2); }
  __device__ inline u32x4 load(ref o) const { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0)); }
};
template <> struct Win<false> {
  typedef const char* ref;
  const char* dummy;
  __device__ inline Win(const McArgs& a, int slot) : dummy(reinterpret_cast<const char*>(a.tmv[slot])) {}
  __device__ inline ref nowhere() const { return dummy; }
  __device__ inline ref plane(const McArgs& a, int pic) const {
    const uint32_t m = pic & 32 ? a.sao_mask_hi : a.sao_mask_lo;
    return a.slab + (size_t)(uint32_t)pic * a.pic_stride + (__builtin_amdgcn_ubfe(m, (uint32_t)pic & 31u, 1u) ? a.sao_off : 0u) + a.origin_off;
  }
  __device__ inline ref at(ref pl, int row, int pitch, int x) const { return pl + ((ptrdiff_t)row * pitch + x) * 2; }
  __device__ inline u32x4 load(ref p) const { return ldg4_a4(p); }
};

// second piece of a window row from the lane to the right (its first piece) in the lanes where `take` holds: a DPP move (row_shl:1 =
// the value of lane + 1; lanes without one, the last of a row of 16, read 0) that the compiler folds into the select (v_cndmask_b32_dpp)
__device__ inline uint32_t from_right(uint32_t own, uint32_t first, bool take) {
  const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)first, 0x101, 0xf, 0xf, true);
  return take ? t : own;
}
__device__ inline u32x4 from_right(const u32x4 own, const u32x4 first, bool take) {
  return u32x4{from_right(own.x, first.x, take), from_right(own.y, first.y, take), from_right(own.z, first.z, take), from_right(own.w, first.w, take)};
}

// ======================================================================================================== luma
// H pass work item: window rows (2i, 2i+1) x 8 columns.  r[0..1] = first row (8 dwords = 16 samples), r[2..3] = second row;
// tap = the item's tap table entry.  out (LDS): piece c at out[c * 64] = (row 2i | row 2i+1 << 16) of columns 2c, 2c+1
__device__ inline void h_item_luma(const u32x4 (&r)[4], const uint32_t* __restrict__ tap, int sh1, uint32_t* __restrict__ out) {
  const u32x4 t0 = *reinterpret_cast<const u32x4*>(tap), t1 = *reinterpret_cast<const u32x4*>(tap + 4);
  const u32x2 t2 = *reinterpret_cast<const u32x2*>(tap + 8);
  const uint32_t te[5] = {t0.x, t0.y, t0.z, t0.w, t1.x}, to[5] = {t1.y, t1.z, t1.w, t2.x, t2.y};
  int sum[2][8];
#pragma unroll
  for (int row = 0; row < 2; row++) {
    const uint32_t d[8] = {r[2 * row].x, r[2 * row].y, r[2 * row].z, r[2 * row].w, r[2 * row + 1].x, r[2 * row + 1].y, r[2 * row + 1].z, r[2 * row + 1].w};
#pragma unroll
    for (int x = 0; x < 8; x++) {
      int v = dot2_first(d[x >> 1], (x & 1) ? to[0] : te[0]);
#pragma unroll
      for (int j = 1; j < 5; j++) v = dot2(d[(x >> 1) + j], (x & 1) ? to[j] : te[j], v);
      sum[row][x] = v;
    }
  }
  // HM: filter<N,false,true,false> (+ 8192): sum >> (6 - headroom), a 16-bit Pel
#pragma unroll
  for (int c = 0; c < 4; c++)
    *reinterpret_cast<u32x2*>(out + c * 64) = u32x2{pack_shr(sum[0][2 * c], sum[1][2 * c], sh1), pack_shr(sum[0][2 * c + 1], sum[1][2 * c + 1], sh1)};
}

// one window row x 8 columns (the halo items of a strip whose second tile row holds no run top are shared out row by row: k_mc_luma)
__device__ inline void h_row_luma(const u32x4 (&r)[2], const uint32_t* __restrict__ tap, int (&sum)[8]) {
  const u32x4 t0 = *reinterpret_cast<const u32x4*>(tap), t1 = *reinterpret_cast<const u32x4*>(tap + 4);
  const u32x2 t2 = *reinterpret_cast<const u32x2*>(tap + 8);
  const uint32_t te[5] = {t0.x, t0.y, t0.z, t0.w, t1.x}, to[5] = {t1.y, t1.z, t1.w, t2.x, t2.y};
  const uint32_t d[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
#pragma unroll
  for (int x = 0; x < 8; x++) {
    int v = dot2_first(d[x >> 1], (x & 1) ? to[0] : te[0]);
#pragma unroll
    for (int j = 1; j < 5; j++) v = dot2(d[(x >> 1) + j], (x & 1) ? to[j] : te[j], v);
    sum[x] = v;
  }
}
// the value lane + 32 holds, in the lanes of the lower half (v_permlane32_swap)
__device__ inline int from_upper_half(int v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1]; }
__device__ inline uint32_t low_half(uint32_t v, int lane) { return (uint32_t)__shfl((int)v, lane & 31); }
__device__ inline const char* low_half(const char* p, int lane) {
  const uint64_t v = (uint64_t)(uintptr_t)p;
  return (const char*)(uintptr_t)(((uint64_t)low_half((uint32_t)(v >> 32), lane) << 32) | low_half((uint32_t)v, lane));
}

// One wave = one strip of the square: tile rows 2w, 2w+1.  Lane -> tile (tx = lane & 7, strip row r = lane >> 5), row pair
// q = (lane >> 3) & 3: the lane's H items are window rows 8+2q, 9+2q ("body") and, if its tile starts a run, rows 2q, 2q+1
// ("halo") of THAT tile; its V item is output rows 2q, 2q+1 of the same tile.  The first tile row of a strip always starts a run
// (the strip above belongs to another wave), so the waves of a workgroup never wait for each other: no barrier, no shared state
// but the tap table, whose copies are identical.
//
// What is NOT loaded (round 3: the kernel is bound by the reference lines its waves pull into L1):
//   * no interpolation in a direction (a quarter of the vectors each): only the 8 rows / 8 columns of the block itself.  Vertical:
//     the window starts one row higher (identity at tap 4 instead of 3), so the block's rows are window rows 4..11 -- the second half
//     of the halo item's rows and the first half of the body item's; the tile shares nothing with its neighbours and is its own run.
//     Horizontal: the window starts at the block's first column (identity at tap 0): one 16-byte piece per row if that column is even.
//   * (experiment MC_DEDUP) the second 16-byte piece of a row when the tile to the right has the same motion and loads the same rows:
//     it IS that lane's first piece (from_right).
//   * window row 15 of a tile that no tile of the strip continues (a tile needs rows 0..14, row 15 is the first row of the next tile's share).
// Rows and pieces that are not loaded only ever meet zero taps.
#ifndef MC_NO_F0
#define MC_F0 1
#endif
// (MC_DEDUP, off: see DESIGN.md -- the move + select per dword cost the arithmetic-bound large PUs what the loads saved)
#ifndef MC_NO_R15
#define MC_R15 1
#endif
template <bool WP, bool BI, bool BUF>
__global__ void __launch_bounds__(256, (WP || BI) ? 1 : MC_LB_LUMA) k_mc_luma(const McArgs a) {
  __shared__ __attribute__((aligned(16))) McLds<false> S;
  int slot; Square g;
  if (!square_of_block(a, slot, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave * 16 >= g.ext) return;
  const int pitch = a.pitch, bd = a.bd;
  const int tx = lane & 7, q = (lane >> 3) & 3, r = lane >> 5;
  const int x0 = g.sx + tx * 8, y0 = g.sy + wave * 16 + r * 8;
  // ---- prologue: the lane's tile record, the tap table
  const u32x4 tm = load_tile_rec(a, slot, g, x0, y0);
  if (lane < 36) *reinterpret_cast<u32x4*>(&S.taps[4 * lane]) = ldg4(&g_taps_luma.e[0][0] + 4 * lane);
  const uint32_t flags = tm.z >> 24, rmask = (tm.w >> 8) & 0xff;
  const bool active = (flags & TM_ACTIVE) != 0;
  if (!__ballot(active)) return;
#if defined(MC_EXP) && (MC_EXP & 64)        // experiment (wrong samples): what sharing rows ACROSS strips would save -- a first-row tile whose
  // upper neighbour in the strip above has the same record does not start a run (its upper rows are read from wherever)
  bool top = tile_is_top(tm, r, lane ^ 32);
  if (wave > 0) {
    const u32x4 up = load_tile_rec(a, slot, g, x0, y0 - 8);
    if (r == 0 && up.x == tm.x && up.y == tm.y && up.z == tm.z) top = false;
  }
#else
  const bool top = tile_is_top(tm, r, lane ^ 32);
#endif
#ifdef MC_DEDUP
  const bool same_right = tx < 7 && !tile_is_top(tm, 1, lane + 1);          // the tile to the right: same lists, pictures, vectors
#else
  const bool same_right = false;
#endif
  const bool any_bi = BI && __ballot((flags & TM_BI) != 0) != 0;
  wave_lds_sync();
#ifdef MC_LDS_PAD
  if (a.n < 0) S.pad[tid] = 1;                               // (keeps the padding allocated)
#endif
  const int npass = any_bi ? 2 : 1;
  const int head = bd >= 12 ? 2 : 14 - bd;
  const int sh1 = 6 - head;
  uint32_t* const body_t = &S.w[wave].body[r * 256 + tx * 2];              // (r, piece 0, row pair 0, tx)
  uint32_t* const halo_t = &S.w[wave].halo[r * 256 + tx * 2];
  const Win<BUF> win(a, slot);
  typedef typename Win<BUF>::ref wref;
  const wref nowhere = win.nowhere();                      // what lanes without a work item load from
  WpTile wp = {false, 1, 0, 1, 0, 0};
  if constexpr (WP) wp = wp_tile<WP>(a.slices[slot], flags, tm.w, 0);
  // the residual of output rows 2q, 2q+1 (two 16-byte slots of the tile's line in PicDev::resid; a row crosses two 4x4 quadrants, each
  // with a residual or without), requested with the tile record in hand.  Lanes without one load the buffer's first bytes.
  uint32_t rsd[2][4];
  {
    const uint32_t qm = active ? (rmask >> (q & 2)) & 3 : 0u;
    // (buffer loads: lanes whose tile carries no residual ask past the end of the buffer and get zeros without a memory access)
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid[slot], 0, kResidLimit, 0x00020000);
    const uint32_t ro = qm ? (uint32_t)((((y0 >> 3) * a.rtw + (x0 >> 3)) * 64 + q * 8) * 2) : kResidLimit;
    const u32x4 r0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro, 0, 0));
    const u32x4 r1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro + 64, 0, 0));
    rsd[0][0] = qm & 1 ? r0.x : 0u; rsd[0][1] = qm & 1 ? r0.y : 0u; rsd[1][0] = qm & 1 ? r1.x : 0u; rsd[1][1] = qm & 1 ? r1.y : 0u;
    rsd[0][2] = qm & 2 ? r0.z : 0u; rsd[0][3] = qm & 2 ? r0.w : 0u; rsd[1][2] = qm & 2 ? r1.z : 0u; rsd[1][3] = qm & 2 ? r1.w : 0u;
  }
  uint32_t park[8];
#pragma unroll
  for (int x = 0; x < 8; x++) park[x] = 0;
#pragma unroll 1
  for (int pass = 0; pass < npass; pass++) {
    const bool bact = active && (pass == 0 || (flags & TM_BI));
    const uint32_t mv = pass ? tm.y : tm.x;
    const uint32_t fr = (tm.z >> (4 * pass)) & 15;
#ifdef MC_F0
    const bool yz = (fr & 12) == 0, xz = (fr & 3) == 0;
#else
    const bool yz = false, xz = false;
#endif
    const bool ptop = top || yz;
    const bool bon = bact && (!yz || q < 2), hon = bact && ptop && (!yz || q >= 2);
    const int xs = x0 + (int)(int16_t)(mv & 0xffff) - (xz ? 0 : 3);
    const bool two = !xz || (xs & 1);
    const uint32_t* const tap = &S.taps[((xz ? 4u : (fr & 3)) * 2 + (xs & 1)) * 12];
    // window row 2q of the tile (the halo item's first row; the body item's is 8 rows down)
    const wref pw = win.at(win.plane(a, (int)((tm.z >> (8 + 8 * pass)) & (kMaxPics - 1))), y0 + ((int)mv >> 16) - (yz ? 4 : 3) + 2 * q, pitch, xs & ~1);
    const uint32_t vidx = yz ? 5u : (fr >> 2) & 3;
    const uint64_t bon_m = __ballot(bon), hon_m = __ballot(hon);
    // the lane to the right loads the same rows of the same window, 16 bytes on: its first piece is this lane's second
#ifdef MC_R15
    // window row 15 (the body item's second row in lanes with q == 3) is only ever read by a tile of the second strip row that continues this one
    const bool b2on = bon && (q != 3 || (r == 0 && !((__ballot(ptop) >> (lane + 32)) & 1)));
#else
    const bool b2on = bon;
#endif
    const uint64_t b2on_m = __ballot(b2on);
    const bool rb_shared = same_right && ((bon_m >> (lane + 1)) & 1), rb2_shared = same_right && ((b2on_m >> (lane + 1)) & 1);
    const bool rh_shared = same_right && ((hon_m >> (lane + 1)) & 1);
    // the H pass of the lane's body item and, in waves that hold run tops, of its halo item.  Loads are unconditional (lanes without
    // an item read nowhere): values that are only defined in some lanes would have to be initialised in the others
    auto h_phase = [&](auto with_halo) {
      constexpr bool HALO = decltype(with_halo)::value;
      u32x4 rb[4], rh[4];
      {
        const wref p0 = bon ? pw + pitch * 16 : nowhere, p1 = bon && two && !rb_shared ? p0 + 16 : nowhere;
        rb[0] = win.load(p0); rb[1] = win.load(p1);
        rb[2] = win.load(b2on ? p0 + pitch * 2 : nowhere); rb[3] = win.load(b2on && two && !rb2_shared ? p0 + pitch * 2 + 16 : nowhere);
      }
      if constexpr (HALO) {
        const wref p0 = hon ? pw : nowhere, p1 = hon && two && !rh_shared ? p0 + 16 : nowhere;
        rh[0] = win.load(p0); rh[1] = win.load(p1);
        rh[2] = win.load(hon ? p0 + pitch * 2 : nowhere); rh[3] = win.load(hon && two && !rh_shared ? p0 + pitch * 2 + 16 : nowhere);
      }
#ifdef MC_DEDUP
      {
        rb[1] = from_right(rb[1], rb[0], rb_shared); rb[3] = from_right(rb[3], rb[2], rb2_shared);
      }
      if constexpr (HALO) {
        rh[1] = from_right(rh[1], rh[0], rh_shared); rh[3] = from_right(rh[3], rh[2], rh_shared);
      }
#endif
      if (bon) h_item_luma(rb, tap, sh1, body_t + q * 16);
      if constexpr (HALO) { if (hon) h_item_luma(rh, tap, sh1, halo_t + q * 16); }
    };
#ifndef MC_NO_SPLIT
    // Run tops in the strip's first tile row only (any PU 16 rows tall or more): their halo items would keep half of the lanes busy with
    // two rows each.  Shared out instead: the lane below (lane + 32: same tile column, same row pair) takes the item's second row --
    // half the arithmetic and two window loads instead of four per lane -- and hands its eight sums up (v_permlane32_swap).
    auto h_phase_split = [&]() {
      u32x4 rb[4], rh[2];
      {
        const wref p0 = bon ? pw + pitch * 16 : nowhere;
        rb[0] = win.load(p0); rb[1] = win.load(bon && two ? p0 + 16 : nowhere);
        rb[2] = win.load(b2on ? p0 + pitch * 2 : nowhere); rb[3] = win.load(b2on && two ? p0 + pitch * 2 + 16 : nowhere);
      }
      const bool hon0 = (hon_m >> (lane & 31)) & 1, two0 = (__ballot(two) >> (lane & 31)) & 1;
      const wref pw0 = low_half(pw, lane) + (r ? pitch * 2 : 0);
      const uint32_t* const tap0 = S.taps + low_half((uint32_t)(tap - S.taps), lane);
      rh[0] = win.load(hon0 ? pw0 : nowhere); rh[1] = win.load(hon0 && two0 ? pw0 + 16 : nowhere);
      if (bon) h_item_luma(rb, tap, sh1, body_t + q * 16);
      int sum[8], up[8];
      h_row_luma(rh, tap0, sum);
#pragma unroll
      for (int x = 0; x < 8; x++) up[x] = from_upper_half(sum[x]);
      if (hon) {                                             // (lanes of the first tile row: hon is never set below it here)
#pragma unroll
        for (int c = 0; c < 4; c++)
          *reinterpret_cast<u32x2*>(halo_t + q * 16 + c * 64) = u32x2{pack_shr(sum[2 * c], up[2 * c], sh1), pack_shr(sum[2 * c + 1], up[2 * c + 1], sh1)};
      }
    };
    if (hon_m && !(hon_m >> 32)) h_phase_split(); else
#endif
    if (hon_m) h_phase(std::true_type()); else h_phase(std::false_type());
    wave_lds_sync();
    if (bact) {
      // output rows 2q, 2q+1 of the tile: window row pairs q .. q+4, the first four of a tile's eight pairs belong to the tile above
      const uint32_t* above = ptop ? halo_t : body_t - 256;
      const uint32_t* vtap = &S.taps[vidx * 2 * 12];
      const u32x4 t0 = *reinterpret_cast<const u32x4*>(vtap), t1 = *reinterpret_cast<const u32x4*>(vtap + 4);
      const u32x2 t2v = *reinterpret_cast<const u32x2*>(vtap + 8);
      const uint32_t A[4] = {t0.x, t0.y, t0.z, t0.w}, B[5] = {t1.y, t1.z, t1.w, t2v.x, t2v.y};
      const int seed = (!WP && !BI) ? 32 << head : 0;       // uni-prediction without weights: the final rounding constant
      int v6[2][8];
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const int gp = q + j;
        const uint32_t* src = gp < 4 ? above + gp * 16 : body_t + (gp - 4) * 16;
        uint32_t pr[8];
#pragma unroll
        for (int c = 0; c < 4; c++) { const u32x2 v = *reinterpret_cast<const u32x2*>(src + c * 64); pr[2 * c] = v.x; pr[2 * c + 1] = v.y; }
#pragma unroll
        for (int x = 0; x < 8; x++) {
          if (j == 0) { v6[0][x] = dot2_seed(pr[x], A[0], seed); v6[1][x] = dot2_seed(pr[x], B[0], seed); }
          else {
            if (j < 4) v6[0][x] = dot2(pr[x], A[j], v6[0][x]);
            v6[1][x] = dot2(pr[x], B[j], v6[1][x]);
          }
        }
      }
      finish_rows<8, WP, BI>(v6, park, pass, (flags & TM_BI) != 0, bd, wp, rsd, a.dst[slot] + (ptrdiff_t)(y0 + 2 * q) * pitch + x0, pitch);
    }
    if (BI && pass + 1 < npass) wave_lds_sync();           // the next list's H pass overwrites what this V pass reads
  }
}

// ====================================================================================================== chroma
// Cb and Cr alternate in ONE plane (hmgpu_dev.h "chroma planes"): a dword of a window row is the pair (Cb, Cr) of one position, so the
// horizontal pass runs on both components at once in packed 16-bit arithmetic, tap by tap (v_pk_mad_i16 with the tap in both halves) --
// no pairing of neighbouring samples, no parity of the window start.  16 bits are enough because every chroma tap is even: with
// S = sum of (c_k / 2) x_k, HM's first-stage value (sum of c_k x_k - (8192 << s)) >> s, s = bit depth - 8 (filter<4, false, true, false>,
// TComInterpolationFilter.cpp:195-212) is ((S - (4096 << s)) << 1) >> s, and S - (4096 << s) lies in [-21499, 21467] at 10 bits, [-5371,
// 5339] at 8 (wrapping on the way is harmless, the end value fits).  Bit depths above 10 take the 32-bit form (h_item_chroma).
struct alignas(16) HTapsChroma { uint32_t e[9][4]; };
constexpr HTapsChroma make_htaps_chroma() {
  constexpr int c[9][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2},
                           {64, 0, 0, 0}};
  HTapsChroma t = {};
  for (int f = 0; f < 9; f++)
    for (int k = 0; k < 4; k++) t.e[f][k] = pk16(c[f][k] / 2, c[f][k] / 2);
  return t;
}
__device__ const HTapsChroma g_htaps_chroma = make_htaps_chroma();
__device__ inline uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {                 // per half: a * b + c (wrapping)
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ inline uint32_t pk_ashr(uint32_t v, uint32_t sh2) {                          // per half: v >> sh (sh2 = sh | sh << 16)
  uint32_t d;
  asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "v"(sh2), "v"(v));
  return d;
}
__device__ inline uint32_t pk_shl(uint32_t v, uint32_t sh2) {
  uint32_t d;
  asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "v"(sh2), "v"(v));
  return d;
}
// H pass work item: window rows (2i, 2i+1) x 4 columns of BOTH components.  r[2 * row], r[2 * row + 1] = the row's 8 (Cb, Cr) pairs from the
// window's first position on.  tap: the fraction's four (c/2, c/2) pairs.  out (LDS): component p, piece c at out[p * 64 + c * 32], a dword =
// (row 2i | row 2i+1 << 16) of one column -- HM's first-stage values WITH their -8192 offset (the V pass seeds its sums with 8192 * 64)
__device__ inline void h_item_chroma(const u32x4 (&r)[4], const uint32_t* __restrict__ tap, int sh1, uint32_t* __restrict__ out) {
  const u32x4 t = *reinterpret_cast<const u32x4*>(tap);
  const uint32_t tc[4] = {t.x, t.y, t.z, t.w};
  if (sh1 > 2) {
    // bit depths above 10 (12: S - (4096 << 4) needs 18 bits): the 32-bit form, one v_dot2 per tap and component -- (c, 0) picks Cb out of a
    // pair, (0, c) Cr; the taps are the table's halves doubled
    const int seed = -(8192 << sh1);
    uint32_t bcb[4], bcr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { bcb[k] = (tc[k] << 1) & 0xffffu; bcr[k] = (tc[k] & 0xffff0000u) << 1; }
    uint32_t hb[2][4], hr[2][4];
#pragma unroll
    for (int row = 0; row < 2; row++) {
      const uint32_t d[8] = {r[2 * row].x, r[2 * row].y, r[2 * row].z, r[2 * row].w, r[2 * row + 1].x, r[2 * row + 1].y, r[2 * row + 1].z, r[2 * row + 1].w};
#pragma unroll
      for (int x = 0; x < 4; x++) {
        int vb = seed, vr = seed;
#pragma unroll
        for (int k = 0; k < 4; k++) { vb = dot2(d[x + k], bcb[k], vb); vr = dot2(d[x + k], bcr[k], vr); }
        hb[row][x] = (uint32_t)(vb >> sh1); hr[row][x] = (uint32_t)(vr >> sh1);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      *reinterpret_cast<u32x2*>(out + c * 32) = u32x2{__builtin_amdgcn_perm(hb[1][2 * c], hb[0][2 * c], 0x05040100u), __builtin_amdgcn_perm(hb[1][2 * c + 1], hb[0][2 * c + 1], 0x05040100u)};
      *reinterpret_cast<u32x2*>(out + 64 + c * 32) = u32x2{__builtin_amdgcn_perm(hr[1][2 * c], hr[0][2 * c], 0x05040100u), __builtin_amdgcn_perm(hr[1][2 * c + 1], hr[0][2 * c + 1], 0x05040100u)};
    }
    return;
  }
  const uint32_t init = (uint32_t)((-(4096 << sh1)) & 0xffff) * 0x10001u;
  uint32_t h[2][4];
#pragma unroll
  for (int row = 0; row < 2; row++) {
    const uint32_t d[8] = {r[2 * row].x, r[2 * row].y, r[2 * row].z, r[2 * row].w, r[2 * row + 1].x, r[2 * row + 1].y, r[2 * row + 1].z, r[2 * row + 1].w};
#pragma unroll
    for (int x = 0; x < 4; x++) {
      uint32_t v = pk_mad(d[x], tc[0], init);
#pragma unroll
      for (int k = 1; k < 4; k++) v = pk_mad(d[x + k], tc[k], v);
      h[row][x] = sh1 == 0 ? pk_shl(v, 0x00010001u) : pk_ashr(v, (uint32_t)(sh1 - 1) * 0x10001u);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    *reinterpret_cast<u32x2*>(out + c * 32) = u32x2{__builtin_amdgcn_perm(h[1][2 * c], h[0][2 * c], 0x05040100u), __builtin_amdgcn_perm(h[1][2 * c + 1], h[0][2 * c + 1], 0x05040100u)};
    *reinterpret_cast<u32x2*>(out + 64 + c * 32) = u32x2{__builtin_amdgcn_perm(h[1][2 * c], h[0][2 * c], 0x07060302u), __builtin_amdgcn_perm(h[1][2 * c + 1], h[0][2 * c + 1], 0x07060302u)};
  }
}

// One wave = one strip (tile rows 2w, 2w+1) as in k_mc_luma.  Lane -> tile (tx = lane & 7, strip row r = (lane >> 4) & 1) for both
// passes.  H item: lanes 0..31 window rows 4+2hq, 5+2hq ("body"), lanes 32..63 rows 2hq, 2hq+1 ("halo", tiles that start a run) of
// both planes, hq = (lane >> 3) & 1.  V item: plane vp = lane >> 5, output rows 2k, 2k+1 with k = (lane >> 3) & 1.
template <bool WP, bool BI, bool BUF>
__global__ void __launch_bounds__(256, (WP && BI) ? 6 : MC_LB_CHROMA) k_mc_chroma(const McArgs a) {
  __shared__ __attribute__((aligned(16))) McLds<true> S;
  int slot; Square g;
  if (!square_of_block(a, slot, g)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave * 16 >= g.ext) return;
  const int pitch = a.pitch, bd = a.bd;
  const int tx = lane & 7, hq = (lane >> 3) & 1, r = (lane >> 4) & 1, hi = lane >> 5;     // hi: H pass halo item / V pass plane
  const int x0 = g.sx + tx * 8, y0 = g.sy + wave * 16 + r * 8;                            // luma position of the tile
  const u32x4 tm = load_tile_rec(a, slot, g, x0, y0);
  if (lane < 36) *reinterpret_cast<u32x4*>(&S.taps[4 * lane]) = ldg4(&g_taps_chroma.e[0][0] + 4 * lane);
  else if (lane < 45) *reinterpret_cast<u32x4*>(&S.htap[4 * (lane - 36)]) = ldg4(&g_htaps_chroma.e[0][0] + 4 * (lane - 36));
  const uint32_t flags = tm.z >> 24, rmask = (tm.w >> 8) & 0xff;
  const bool active = (flags & TM_ACTIVE) != 0;
  if (!__ballot(active)) return;
  const bool top = tile_is_top(tm, r, lane ^ 16);
  const bool any_bi = BI && __ballot((flags & TM_BI) != 0) != 0;
  wave_lds_sync();
  const int npass = any_bi ? 2 : 1;
  const int head = bd >= 12 ? 2 : 14 - bd;
  const int sh1 = 6 - head;
  uint32_t* const hout = (hi ? S.w[wave].halo : S.w[wave].body) + r * 128 + tx * 2 + hq * 16;     // (r, plane 0, piece 0, hq, tx)
  uint32_t* const body_v = &S.w[wave].body[r * 128 + hi * 64 + tx * 2];                           // (r, vp, piece 0, row pair 0, tx)
  uint32_t* const halo_v = &S.w[wave].halo[r * 128 + hi * 64 + tx * 2];
  WpTile wp = {false, 1, 0, 1, 0, 0};
  if constexpr (WP) wp = wp_tile<WP>(a.slices[slot], flags, tm.w, 1 + hi);
  // the residual of the plane's rows 2hq, 2hq+1 of the 4x4 block under the tile (8 bytes each, in the 8x8 chroma tile of
  // PicDev::resid), requested with the tile record in hand: its latency hides behind the whole H pass
  uint32_t rsd[2][2];
  {
    const bool coded = active && (rmask & (hi ? TR_CR : TR_CB));
    const int xc = x0 >> 1, yc = (y0 >> 1) + 2 * hq;
    // (both planes' residual tiles lie in one allocation, Cb before Cr: one buffer, the plane is part of the offset)
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid[slot], 0, kResidLimit, 0x00020000);
    const uint32_t ro = coded ? (uint32_t)(((((yc >> 3) * a.rtw + (xc >> 3)) * 8 + resid_slot(yc)) * 8 + (xc & 4)) * 2) + (hi ? (uint32_t)((const char*)a.resid2[slot] - (const char*)a.resid[slot]) : 0u) : kResidLimit;
    const u32x2 r0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, ro, 0, 0));
    const u32x2 r1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, ro + 64, 0, 0));
    rsd[0][0] = coded ? r0.x : 0u; rsd[0][1] = coded ? r0.y : 0u; rsd[1][0] = coded ? r1.x : 0u; rsd[1][1] = coded ? r1.y : 0u;
  }
  uint32_t park[4] = {0, 0, 0, 0};
  const Win<BUF> win(a, slot);
  typedef typename Win<BUF>::ref wref;
#pragma unroll 1
  for (int pass = 0; pass < npass; pass++) {
    const bool vact = active && (pass == 0 || (flags & TM_BI));
    const uint32_t mv = pass ? tm.y : tm.x, fr = (tm.z >> (4 * pass)) & 15;
    const int ix = (int)(int16_t)(mv & 0xffff), iy = (int)mv >> 16;
    // chroma vector = luma vector in eighth samples: integer part ix >> 1, fraction (ix & 1) * 4 + quarter fraction
    const int yf = ((iy & 1) << 2) | ((fr >> 2) & 3);
#ifdef MC_F0
    // no vertical interpolation (an eighth of the vectors): the window starts one row higher (identity at tap 2), the block's four rows are
    // window rows 2..5 -- the halo item's second row pair and the body item's first; the tile is its own run (as in k_mc_luma)
    const bool yz = yf == 0;
#else
    const bool yz = false;
#endif
    const bool ptop = top || yz;
    const bool hact = vact && (hi == 0 ? (!yz || hq == 0) : (ptop && (!yz || hq == 1)));
#ifdef MC_R15
    // window row 7 (the second row of the body item with hq == 1) is only ever read by a tile of the second strip row that continues this one
    const bool row2 = hi != 0 || hq == 0 || (r == 0 && !((__ballot(ptop) >> (lane | 16)) & 1));
#else
    const bool row2 = true;
#endif
    if (hact) {
      const int xf = ((ix & 1) << 2) | (fr & 3);
#ifdef MC_F0
      const bool xz = xf == 0;          // no horizontal interpolation: the window starts at the block's first column (identity at tap 0), one piece per row
#else
      const bool xz = false;
#endif
      const int xs = (x0 >> 1) + (ix >> 1) - (xz ? 0 : 1), ys = (y0 >> 1) + (iy >> 1) - (yz ? 2 : 1) + (hi ? 0 : 4) + 2 * hq;
      // a window row = the (Cb, Cr) pairs xs .. xs + 6 of one plane row: two 16-byte pieces of ONE line (hmgpu_dev.h "chroma planes")
      const wref p0 = win.at(win.plane(a, (int)((tm.z >> (8 + 8 * pass)) & (kMaxPics - 1))), ys, pitch, kCStep * xs);
      const int rowb = pitch * 2;
      u32x4 rr[4];
#if defined(MC_CEXP) && (MC_CEXP & 1)       // experiment: no halo rows
      const bool ld = hi == 0;
#elif defined(MC_CEXP) && (MC_CEXP & 2)     // experiment: no window loads at all
      const bool ld = false;
#else
      const bool ld = true;
#endif
      rr[0] = win.load(ld ? p0 : win.nowhere()); rr[1] = win.load(ld && !xz ? p0 + 16 : win.nowhere());
      rr[2] = win.load(ld && row2 ? p0 + rowb : win.nowhere()); rr[3] = win.load(ld && row2 && !xz ? p0 + rowb + 16 : win.nowhere());
      h_item_chroma(rr, &S.htap[(xz ? 8 : xf) * 4], sh1, hout);
    }
    wave_lds_sync();
    if (vact) {
      const uint32_t* above = ptop ? halo_v : body_v - 128;
      const uint32_t* vtap = &S.taps[(yz ? 8 : yf) * 2 * 8];
      const u32x4 t0 = *reinterpret_cast<const u32x4*>(vtap);
      const u32x2 t1 = *reinterpret_cast<const u32x2*>(vtap + 4);
      const uint32_t A[2] = {t0.x, t0.y}, B[3] = {t0.w, t1.x, t1.y};
      // (the first-stage values in LDS carry HM's -8192: the taps sum to 64, so 8192 * 64 brings the sums to what finish_rows expects)
      const int seed = (8192 << 6) + ((!WP && !BI) ? 32 << head : 0);
      int v6[2][4];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int gp = hq + j;
        const uint32_t* src = gp < 2 ? above + gp * 16 : body_v + (gp - 2) * 16;
        const u32x2 c0 = *reinterpret_cast<const u32x2*>(src), c1 = *reinterpret_cast<const u32x2*>(src + 32);
        const uint32_t pr[4] = {c0.x, c0.y, c1.x, c1.y};
#pragma unroll
        for (int x = 0; x < 4; x++) {
          if (j == 0) { v6[0][x] = dot2_seed(pr[x], A[0], seed); v6[1][x] = dot2_seed(pr[x], B[0], seed); }
          else {
            if (j < 2) v6[0][x] = dot2(pr[x], A[j], v6[0][x]);
            v6[1][x] = dot2(pr[x], B[j], v6[1][x]);
          }
        }
      }
      const int xc = x0 >> 1, yc = (y0 >> 1) + 2 * hq;
      // the lane holds rows 2hq, 2hq+1 x 4 columns of ONE component; the plane wants (Cb, Cr) pairs: the two lanes of a position (32 apart)
      // swap a row each and store 16 bytes -- lane hi writes row 2hq + hi of both components (finish_rows, IL)
      finish_rows<4, WP, BI, true>(v6, park, pass, (flags & TM_BI) != 0, bd, wp, rsd, a.dst[slot] + (ptrdiff_t)(yc + hi) * pitch + kCStep * xc, pitch);
    }
    if (BI && pass + 1 < npass) wave_lds_sync();
  }
}

// grid: the blocks of one picture run on the XCDs x with x % n == picture (each XCD has its own L2; workgroups are dealt
// round-robin to the 8 XCDs in dispatch order, x fastest), walking the picture's squares in raster order: the order of
// xcd_remap(), without its divisions.  mode 1: blockIdx.x = XCD, blockIdx.y = position inside the XCD's band; mode 0: blockIdx.x = picture
template <typename K>
static void launch_mc(K kernel, McArgs& a, int max_ctus, hipStream_t s) {
  a.mode = 0; a.log2n = 0; a.per = 0;
  dim3 grid((unsigned)a.n, (unsigned)max_ctus);
  if (a.n <= 8 && (8 % a.n) == 0) {
    const int m = 8 / a.n;
    a.mode = 1; a.per = (max_ctus + m - 1) / m;
    while ((1 << a.log2n) < a.n) a.log2n++;
    grid = dim3(8u, (unsigned)a.per);
  }
  hipLaunchKernelGGL(kernel, grid, dim3(256), 0, s, a);
}
// buffer-load windows while every byte of the slab of pictures has a 32-bit offset below kWinLimit (57 pictures of 3840x2160), plain addresses beyond
#ifdef MC_FORCE_PTR
static bool win_buf(const McArgs& a) { return false; }
#else
static bool win_buf(const McArgs& a) { return a.slab_bytes < (uint64_t)kWinLimit; }
#endif
void launch_mc_luma(McArgs& a, int max_ctus, bool wp, bool bi, hipStream_t s) {
  if (win_buf(a)) {
    if (wp) { if (bi) launch_mc(k_mc_luma<true, true, true>, a, max_ctus, s); else launch_mc(k_mc_luma<true, false, true>, a, max_ctus, s); }
    else { if (bi) launch_mc(k_mc_luma<false, true, true>, a, max_ctus, s); else launch_mc(k_mc_luma<false, false, true>, a, max_ctus, s); }
  } else {
    if (wp) { if (bi) launch_mc(k_mc_luma<true, true, false>, a, max_ctus, s); else launch_mc(k_mc_luma<true, false, false>, a, max_ctus, s); }
    else { if (bi) launch_mc(k_mc_luma<false, true, false>, a, max_ctus, s); else launch_mc(k_mc_luma<false, false, false>, a, max_ctus, s); }
  }
}
void launch_mc_chroma(McArgs& a, int max_ctus, bool wp, bool bi, hipStream_t s) {
  if (win_buf(a)) {
    if (wp) { if (bi) launch_mc(k_mc_chroma<true, true, true>, a, max_ctus, s); else launch_mc(k_mc_chroma<true, false, true>, a, max_ctus, s); }
    else { if (bi) launch_mc(k_mc_chroma<false, true, true>, a, max_ctus, s); else launch_mc(k_mc_chroma<false, false, true>, a, max_ctus, s); }
  } else {
    if (wp) { if (bi) launch_mc(k_mc_chroma<true, true, false>, a, max_ctus, s); else launch_mc(k_mc_chroma<true, false, false>, a, max_ctus, s); }
    else { if (bi) launch_mc(k_mc_chroma<false, true, false>, a, max_ctus, s); else launch_mc(k_mc_chroma<false, false, false>, a, max_ctus, s); }
  }
}

}  // namespace hmgpu
