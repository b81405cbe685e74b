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
#define MC_LB_LUMA 1
#endif
#ifndef MC_LB_CHROMA
#define MC_LB_CHROMA 1
#endif

__device__ inline uint32_t pk_sub(uint32_t a, uint32_t b) {                           // per half: a - b (wrapping)
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}

#ifdef MC_STAMP
#define STAMP(i) do { if (stamp_on) { stamp[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#define STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define STAMP(i) do {} while (0)
#define STAMP_DRAIN() do {} while (0)
#endif

constexpr uint32_t TM_TOP = 0x80;          // local: set in the flags byte of a run's first tile

// ---- tap tables: entry (fraction f, parity p) of an N-tap filter c[]: with A = (c0,c1)(c2,c3).. and B = (0,c0)(c1,c2)..(c[N-1],0)
//   p = 0: even columns A,0   odd columns B        p = 1: even columns B   odd columns 0,A
// (column x of a window that starts in the high half of its first dword begins one sample later than its dword)
struct alignas(16) TapsLuma { uint32_t e[8][12]; };
struct alignas(16) TapsChroma { uint32_t e[16][8]; };
constexpr uint32_t pk16(int a, int b) { return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16); }
constexpr TapsLuma make_taps_luma() {
  constexpr int c[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
  TapsLuma t = {};
  for (int f = 0; f < 4; f++)
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
  constexpr int c[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};
  TapsChroma t = {};
  for (int f = 0; f < 8; f++)
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

// LDS: H-pass results as 8-byte pieces (two columns x one row pair), laid out [tile row][piece][row pair][tile column] so that
// every DS instruction of both passes touches each bank once: the 16 / 32 lanes an LDS cycle serves are the eight tiles of a
// tile row times two / four row pairs (ds_write_b64 / ds_read_b64), whichever region (own tile, tile above, halo) a lane reads.
//   luma:   piece c = columns 2c, 2c+1 (c < 4), row pair g < 4:            byte (ty, c, g, tx) = ty * 1024 + c * 256 + g * 64 + tx * 8
//   chroma: plane p, piece c = columns 2c, 2c+1 (c < 2), row pair g < 2:   byte (ty, p, c, g, tx) = ty * 512 + p * 256 + c * 128 + g * 64 + tx * 8
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

template <bool CHROMA> struct McLds {
  static constexpr int ROW_DW = CHROMA ? 128 : 256;           // dwords per tile row
  uint32_t body[8 * ROW_DW];
  uint32_t halo[8 * ROW_DW];                                  // the first 8 (4) window rows of run tops, at the top tile's position
  uint32_t taps[CHROMA ? 128 : 96];                           // the tap table (every wave writes the same values, reads its own)
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
      const int l0 = (flags & TM_FIRST_L1) ? 1 : 0, r0 = wpw & 0xff, r1 = (wpw >> 8) & 0xff;
      w.w0 = ldg(&sd.wp_weight[l0][r0][comp]); w.o0 = ldg(&sd.wp_offset[l0][r0][comp]);
      if (flags & TM_BI) { w.w1 = ldg(&sd.wp_weight[1][r1][comp]); w.o1 = ldg(&sd.wp_offset[1][r1][comp]); }
    }
  }
  return w;
}

// the end of a V-pass thread: v6[r][x] = vertical sums of two output rows WITHOUT HM's intermediate offset: v6 = HM's sum + 8192 * 64.
//   pass 0, uni: final samples (filter isLast / weightUnidir) -> dst;  pass 0, bi: HM's 14-bit values (16-bit Pel) wait in park[];
//   pass 1: addAvg / weightBidir with the parked first list -> dst
template <int W, bool WP, bool BI>
__device__ inline void finish_rows(int (&v6)[2][W], uint32_t (&park)[W], int pass, bool bi, int bd, const WpTile& wp,
                                   int16_t* __restrict__ dst, int pitch) {
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
  for (int r = 0; r < 2; r++) {
    int16_t* row = dst + (ptrdiff_t)r * pitch;
    if constexpr (W == 8) { u32x4 v = {res[r][0], res[r][1], res[r][2], res[r][3]}; stg4(row, v); }
    else { u32x2 v = {res[r][0], res[r][1]}; stg2(row, v); }
  }
}

// ---- what every wave knows after the prologue (lane = tile of the square)
struct WaveTiles {
  uint32_t w0, w1, w2, w3;     // this lane's TileMv (w2: frac | ref0 << 8 | ref1 << 16 | (flags | TM_TOP) << 24)
  int tl;                      // lane r: the r-th run top (raster order)
  int ntop;
  bool any_bi;
};
__device__ inline u32x4 load_tile_rec(const McArgs& a, int slot, const Square& g) {
  const int lane = threadIdx.x & 63, ltx = lane & 7, lty = lane >> 3;
  const bool tin = ltx * 8 < g.ext && lty * 8 < g.ext && g.sx + ltx * 8 < a.width && g.sy + lty * 8 < a.height;
  u32x4 tm = {0, 0, 0, 0};
  if (tin) tm = ldg4(a.tmv[slot] + (size_t)((g.sy >> 3) + lty) * a.tw + (g.sx >> 3) + ltx);
  return tm;
}
__device__ inline bool wave_prologue(const u32x4 tm, WaveTiles& T) {
  const int lane = threadIdx.x & 63;
  const bool active = ((tm.z >> 24) & TM_ACTIVE) != 0;
  const unsigned long long mact = __ballot(active);
  if (!mact) return false;
  // the tile continues the run of the tile above: same lists, pictures, vectors (the windows are then exactly one tile apart)
  const uint32_t u0 = __shfl_up(tm.x, 8), u1 = __shfl_up(tm.y, 8), u2 = __shfl_up(tm.z, 8);
  const bool top = active && !(lane >= 8 && u0 == tm.x && u1 == tm.y && u2 == tm.z);
  const unsigned long long mtop = __ballot(top);
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mtop >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mtop, 0u));
  T.ntop = __popcll(mtop);
  T.tl = __builtin_amdgcn_ds_permute((top ? rank : T.ntop + lane - rank) * 4, lane);
  T.w0 = tm.x; T.w1 = tm.y; T.w2 = tm.z | (top ? TM_TOP << 24 : 0u); T.w3 = tm.w;
  T.any_bi = __ballot(((tm.z >> 24) & TM_BI) != 0) != 0;
  return true;
}
// address of sample (0,0) of the final luma / Cb plane of device picture `ref` (all pictures of a context live in one slab, a
// finished picture is its SAO output or, without SAO, its reconstruction)
__device__ inline const char* final_plane(const McArgs& a, int ref) {
  const uint32_t m = ref & 32 ? a.sao_mask_hi : a.sao_mask_lo;
  return a.slab + (size_t)(uint32_t)ref * a.pic_stride + (__builtin_amdgcn_ubfe(m, (uint32_t)ref & 31u, 1u) ? a.sao_off : 0u) + a.origin_off;
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

template <bool WP, bool BI>
__global__ void __launch_bounds__(256, MC_LB_LUMA) k_mc_luma(const McArgs a) {
  __shared__ __attribute__((aligned(16))) McLds<false> S;
#ifdef MC_STAMP
  unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool stamp_on = a.stamps != nullptr;
  STAMP(0);
#endif
  int slot; Square g;
  if (!square_of_block(a, slot, g)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int pitch = a.pitch, bd = a.bd;
  // ---- prologue: tile records, the tap table
  WaveTiles T;
  const u32x4 tmrec = load_tile_rec(a, slot, g);
  if (lane < 48) *reinterpret_cast<u32x2*>(&S.taps[2 * lane]) = ldg2(&g_taps_luma.e[0][0] + 2 * lane);
  STAMP_DRAIN(); STAMP(1);
  if (!wave_prologue(tmrec, T)) return;
  wave_lds_sync();
#ifdef MC_LDS_PAD
  if (a.n < 0) S.pad[tid] = 1;                               // (keeps the padding allocated)
#endif
  const int npass = (BI && T.any_bi) ? 2 : 1;
  const int head = bd >= 12 ? 2 : 14 - bd;
  const int sh1 = 6 - head;
  // body item / V item of this thread: tile t = (tx, ty), row pair q;  halo item: the t-th run top, row pair q
  const int tx = tid & 7, q = (tid >> 3) & 3, ty = tid >> 5, t = ty * 8 + tx;
  const uint32_t b0 = __shfl((int)T.w0, t), b1 = BI ? __shfl((int)T.w1, t) : 0u, b2 = __shfl((int)T.w2, t);
  const bool has_halo = t < T.ntop;
  const int t2 = __shfl(T.tl, t);
  const uint32_t h0 = __shfl((int)T.w0, t2), h1 = BI ? __shfl((int)T.w1, t2) : 0u, h2 = __shfl((int)T.w2, t2);
  const uint32_t bflags = b2 >> 24, hflags = h2 >> 24;
  WpTile wp = {false, 1, 0, 1, 0, 0};
  if constexpr (WP) wp = wp_tile<WP>(a.slices[slot], bflags, (uint32_t)__shfl((int)T.w3, t), 0);
  const int x0 = g.sx + tx * 8, y0 = g.sy + ty * 8;
  uint32_t* const body_t = &S.body[ty * 256 + tx * 2];                   // (ty, piece 0, row pair 0, tx)
  uint32_t* const halo_t2 = &S.halo[(t2 >> 3) * 256 + (t2 & 7) * 2];
  const char* const dummy = reinterpret_cast<const char*>(a.tmv[slot]);   // where lanes without a work item load from (valid memory)
  uint32_t park[8];
#pragma unroll
  for (int x = 0; x < 8; x++) park[x] = 0;
#pragma unroll 1
  for (int pass = 0; pass < npass; pass++) {
    const bool bact = (bflags & TM_ACTIVE) && (pass == 0 || (bflags & TM_BI));
    const bool hact = has_halo && (pass == 0 || (hflags & TM_BI));
    const uint32_t bmv = pass ? b1 : b0, hmv = pass ? h1 : h0;
    const uint32_t bfr = (b2 >> (4 * pass)) & 15, hfr = (h2 >> (4 * pass)) & 15;
    // the H pass of this thread's body item and, in waves that hold run tops, of its halo item.  Loads are unconditional (lanes
    // without an item read a dummy address): values that are only defined in some lanes would have to be initialised in the others
    auto h_phase = [&](auto with_halo) {
      constexpr bool HALO = decltype(with_halo)::value;
      u32x4 rb[4], rh[4];
      const int bxs = x0 + (int)(int16_t)(bmv & 0xffff) - 3;
      {
        const int ys = y0 + ((int)bmv >> 16) - 3 + 8 + 2 * q;
        const char* p0 = final_plane(a, (int)((b2 >> (8 + 8 * pass)) & (kMaxPics - 1))) + ((ptrdiff_t)ys * pitch + (bxs & ~1)) * 2;
        if (!bact) p0 = dummy;
        rb[0] = ldg4_a4(p0); rb[1] = ldg4_a4(p0 + 16); rb[2] = ldg4_a4(p0 + pitch * 2); rb[3] = ldg4_a4(p0 + pitch * 2 + 16);
      }
      const int hxs = g.sx + (t2 & 7) * 8 + (int)(int16_t)(hmv & 0xffff) - 3;
      if constexpr (HALO) {
        const int ys = g.sy + (t2 >> 3) * 8 + ((int)hmv >> 16) - 3 + 2 * q;
        const char* p0 = final_plane(a, (int)((h2 >> (8 + 8 * pass)) & (kMaxPics - 1))) + ((ptrdiff_t)ys * pitch + (hxs & ~1)) * 2;
        if (!hact) p0 = dummy;
        rh[0] = ldg4_a4(p0); rh[1] = ldg4_a4(p0 + 16); rh[2] = ldg4_a4(p0 + pitch * 2); rh[3] = ldg4_a4(p0 + pitch * 2 + 16);
      }
      STAMP(2); STAMP_DRAIN(); STAMP(3);
      if (bact) h_item_luma(rb, &S.taps[((bfr & 3) * 2 + (bxs & 1)) * 12], sh1, body_t + q * 16);
      if constexpr (HALO) { if (hact) h_item_luma(rh, &S.taps[((hfr & 3) * 2 + (hxs & 1)) * 12], sh1, halo_t2 + q * 16); }
    };
    if (__ballot(hact)) h_phase(std::true_type()); else h_phase(std::false_type());
    STAMP(4);
    __syncthreads();
    STAMP(5);
    if (bact) {
      // output rows 2q, 2q+1 of the tile: window row pairs q .. q+4, the first four of a tile's eight pairs belong to the tile above
      const uint32_t* above = (bflags & TM_TOP) ? &S.halo[ty * 256 + tx * 2] : body_t - 256;
      const uint32_t* tap = &S.taps[((bfr >> 2) & 3) * 2 * 12];
      const u32x4 t0 = *reinterpret_cast<const u32x4*>(tap), t1 = *reinterpret_cast<const u32x4*>(tap + 4);
      const u32x2 t2v = *reinterpret_cast<const u32x2*>(tap + 8);
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
      finish_rows<8, WP, BI>(v6, park, pass, (bflags & TM_BI) != 0, bd, wp, a.dst[slot] + (ptrdiff_t)(y0 + 2 * q) * pitch + x0, pitch);
    }
    if (BI && pass + 1 < npass) __syncthreads();           // the next list's H pass overwrites what this V pass reads
  }
#ifdef MC_STAMP
  STAMP(6); STAMP_DRAIN(); STAMP(7);
  if (stamp_on && lane == 0) {
    const size_t w = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6);
    for (int i = 0; i < 8; i++) a.stamps[w * 8 + i] = stamp[i];
  }
#endif
}

// ====================================================================================================== chroma
// H pass work item: window rows (2i, 2i+1) x 4 columns of BOTH planes.  r[2 * plane + row] = one row (4 dwords = 8 samples).
// out (LDS): plane p, piece c at out[p * 64 + c * 32]
__device__ inline void h_item_chroma(const u32x4 (&r)[4], const uint32_t* __restrict__ tap, int sh1, uint32_t* __restrict__ out) {
  const u32x4 t0 = *reinterpret_cast<const u32x4*>(tap);
  const u32x2 t1 = *reinterpret_cast<const u32x2*>(tap + 4);
  const uint32_t te[3] = {t0.x, t0.y, t0.z}, to[3] = {t0.w, t1.x, t1.y};
#pragma unroll
  for (int pl = 0; pl < 2; pl++) {
    int sum[2][4];
#pragma unroll
    for (int row = 0; row < 2; row++) {
      const uint32_t d[4] = {r[2 * pl + row].x, r[2 * pl + row].y, r[2 * pl + row].z, r[2 * pl + row].w};
#pragma unroll
      for (int x = 0; x < 4; x++) {
        int v = dot2_first(d[x >> 1], (x & 1) ? to[0] : te[0]);
#pragma unroll
        for (int j = 1; j < 3; j++) v = dot2(d[(x >> 1) + j], (x & 1) ? to[j] : te[j], v);
        sum[row][x] = v;
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++)
      *reinterpret_cast<u32x2*>(out + pl * 64 + c * 32) = u32x2{pack_shr(sum[0][2 * c], sum[1][2 * c], sh1), pack_shr(sum[0][2 * c + 1], sum[1][2 * c + 1], sh1)};
  }
}

template <bool WP, bool BI>
__global__ void __launch_bounds__(256, MC_LB_CHROMA) k_mc_chroma(const McArgs a) {
  __shared__ __attribute__((aligned(16))) McLds<true> S;
  int slot; Square g;
  if (!square_of_block(a, slot, g)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int pitch = a.pitch, bd = a.bd;
  WaveTiles T;
  const u32x4 tmrec = load_tile_rec(a, slot, g);
  *reinterpret_cast<u32x2*>(&S.taps[2 * lane]) = ldg2(&g_taps_chroma.e[0][0] + 2 * lane);
  if (!wave_prologue(tmrec, T)) return;
  wave_lds_sync();
  const int npass = (BI && T.any_bi) ? 2 : 1;
  const int head = bd >= 12 ? 2 : 14 - bd;
  const int sh1 = 6 - head;
  // H item: threads 0..127 the body items (tile, row pair hq) of both planes; threads 128..255 the halo items of the run tops
  const int tx = tid & 7, hq = (tid >> 3) & 1;
  const bool is_halo = tid >= 128;
  const int hi = ((tid & 127) >> 4) * 8 + tx;              // body: tile;  halo: index into the run tops
  const bool has_item = !is_halo || hi < T.ntop;
  const int th = is_halo ? __shfl(T.tl, hi) : hi;
  const uint32_t h0 = __shfl((int)T.w0, th), h1 = BI ? __shfl((int)T.w1, th) : 0u, h2 = __shfl((int)T.w2, th);
  const uint32_t hflags = has_item ? h2 >> 24 : 0u;
  // V item: tile t = (tx, ty), plane vp, output rows 2k, 2k+1
  const int k = (tid >> 3) & 1, vp = (tid >> 4) & 1, ty = tid >> 5, t = ty * 8 + tx;
  const uint32_t v0 = __shfl((int)T.w0, t), v1 = BI ? (uint32_t)__shfl((int)T.w1, t) : 0u, v2 = __shfl((int)T.w2, t);
  const uint32_t vflags = v2 >> 24;
  WpTile wp = {false, 1, 0, 1, 0, 0};
  if constexpr (WP) wp = wp_tile<WP>(a.slices[slot], vflags, (uint32_t)__shfl((int)T.w3, t), 1 + vp);
  uint32_t* const hout = (is_halo ? S.halo : S.body) + (th >> 3) * 128 + (th & 7) * 2 + hq * 16;      // (ty, plane 0, piece 0, hq, tx)
  uint32_t* const body_v = &S.body[ty * 128 + vp * 64 + tx * 2];                                   // (ty, vp, piece 0, row pair 0, tx)
  uint32_t park[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int pass = 0; pass < npass; pass++) {
    const bool hact = (hflags & TM_ACTIVE) && (pass == 0 || (hflags & TM_BI));
    const bool vact = (vflags & TM_ACTIVE) && (pass == 0 || (vflags & TM_BI));
    if (hact) {
      const uint32_t mv = pass ? h1 : h0, fr = (h2 >> (4 * pass)) & 15;
      const int ix = (int)(int16_t)(mv & 0xffff), iy = (int)mv >> 16;
      // chroma vector = luma vector in eighth samples: integer part ix >> 1, fraction (ix & 1) * 4 + quarter fraction
      const int xs = ((g.sx >> 1) + (th & 7) * 4) + (ix >> 1) - 1, ys = ((g.sy >> 1) + (th >> 3) * 4) + (iy >> 1) - 1 + (is_halo ? 0 : 4) + 2 * hq;
      const int xf = ((ix & 1) << 2) | (fr & 3);
      const char* p0 = final_plane(a, (int)((h2 >> (8 + 8 * pass)) & (kMaxPics - 1))) + ((ptrdiff_t)ys * pitch + (xs & ~1)) * 2;
      const char* p1 = p0 + a.cr_off;
      u32x4 r[4];
      r[0] = ldg4_a4(p0); r[1] = ldg4_a4(p0 + pitch * 2); r[2] = ldg4_a4(p1); r[3] = ldg4_a4(p1 + pitch * 2);
      h_item_chroma(r, &S.taps[(xf * 2 + (xs & 1)) * 8], sh1, hout);
    }
    __syncthreads();
    if (vact) {
      const int iy = (int)(pass ? v1 : v0) >> 16;
      const int yf = ((iy & 1) << 2) | ((v2 >> (4 * pass + 2)) & 3);
      const uint32_t* above = (vflags & TM_TOP) ? &S.halo[ty * 128 + vp * 64 + tx * 2] : body_v - 128;
      const uint32_t* tap = &S.taps[yf * 2 * 8];
      const u32x4 t0 = *reinterpret_cast<const u32x4*>(tap);
      const u32x2 t1 = *reinterpret_cast<const u32x2*>(tap + 4);
      const uint32_t A[2] = {t0.x, t0.y}, B[3] = {t0.w, t1.x, t1.y};
      const int seed = (!WP && !BI) ? 32 << head : 0;
      int v6[2][4];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int gp = k + j;
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
      const int xc = (g.sx >> 1) + tx * 4, yc = (g.sy >> 1) + ty * 4 + 2 * k;
      finish_rows<4, WP, BI>(v6, park, pass, (vflags & TM_BI) != 0, bd, wp, (vp ? a.dst2[slot] : a.dst[slot]) + (ptrdiff_t)yc * pitch + xc, pitch);
    }
    if (BI && pass + 1 < npass) __syncthreads();
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
void launch_mc_luma(McArgs& a, int max_ctus, bool wp, bool bi, hipStream_t s) {
  if (wp) { if (bi) launch_mc(k_mc_luma<true, true>, a, max_ctus, s); else launch_mc(k_mc_luma<true, false>, a, max_ctus, s); }
  else { if (bi) launch_mc(k_mc_luma<false, true>, a, max_ctus, s); else launch_mc(k_mc_luma<false, false>, a, max_ctus, s); }
}
void launch_mc_chroma(McArgs& a, int max_ctus, bool wp, bool bi, hipStream_t s) {
  if (wp) { if (bi) launch_mc(k_mc_chroma<true, true>, a, max_ctus, s); else launch_mc(k_mc_chroma<true, false>, a, max_ctus, s); }
  else { if (bi) launch_mc(k_mc_chroma<false, true>, a, max_ctus, s); else launch_mc(k_mc_chroma<false, false>, a, max_ctus, s); }
}

}  // namespace hmgpu
