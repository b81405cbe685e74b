// k_intra.hip -- reconstruction of intra CUs: reference samples, smoothing, planar / DC / angular prediction, residual.
//   TDecCu::xReconIntraQT -> xIntraRecQT -> xIntraRecBlk                                   TDecCu.cpp:484-730
//   TComPrediction::initAdiPatternChType, fillReferenceSamples, availability helpers       TComPattern.cpp:107-700
//   TComPrediction::predIntraAng, xPredIntraAng, xPredIntraPlanar, xDCPredFiltering        TComPrediction.cpp:182-491,746-840
//
// Intra prediction is the one serial chain of the reconstruction path: a TU predicts from the reconstructed samples of the
// TUs before it in decoding order.  What IS independent: the three components (HM reconstructs the luma of a CU, then its
// chroma, and chroma never reads luma samples: TDecCu.cpp:665-690), and CTU rows once the CTU above-right is done.  So one
// wave per CTU and component; a CTU starts when those of its left / above-left / above / above-right neighbours that hold intra
// CUs are done (one flag per CTU and component).  CTUs without intra CUs (flags from k_prep) cost nothing: their samples were
// finished by the MC / residual kernels before this kernel started.  Inside a CTU the wave takes the TUs in z order; for each TU
// the lanes build the 4N+1 reference samples together (availability per 4x4 unit as one ballot mask, the substitution
// process of fillReferenceSamples as bit scans over that mask), lane n predicts row n, adds the TU's residual and the clipped row goes
// back to the picture.  The residual itself -- de-quantisation, inverse transform, the RExt rotation / RDPCM -- depends on no
// neighbour: k_prep lists the coded TUs of intra CUs beside the inter ones and k_itx has computed them before this kernel starts
// (round 3; the transform inside the TU chain was 14 % of an I picture's time).
//
// Samples written here are read by other waves on other XCDs (whose L2s are not coherent with each other): every access to
// the picture planes in this kernel is an agent-scope atomic dword access (served at the coherent level), ordered against
// the progress counters by release / acquire.
#include "hmgpu_dev.h"
#include "itx_core.h"       // packed 16-bit helpers, wave_lds_sync
#include <algorithm>
#include <type_traits>

namespace hmgpu {

namespace {


// coherent accesses to the picture being reconstructed
#if defined(INTRA_EXP) && (INTRA_EXP & 64)      // experiment (stale samples possible): plain loads
__device__ inline uint32_t ld_coh(const uint32_t* p) { return *p; }
#else
__device__ inline uint32_t ld_coh(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
#if defined(INTRA_EXP) && (INTRA_EXP & 32)      // experiment (other CTUs may read stale samples): plain stores
__device__ inline void st_coh(uint32_t* p, uint32_t v) { *p = v; }
__device__ inline void st_coh2(uint32_t* p, uint32_t v0, uint32_t v1) { *(unsigned long long*)p = (unsigned long long)v0 | ((unsigned long long)v1 << 32); }
#else
__device__ inline void st_coh(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_coh2(uint32_t* p, uint32_t v0, uint32_t v1) {         // four samples, p 8-byte aligned
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v0 | ((unsigned long long)v1 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
// chroma (hmgpu_dev.h "chroma planes": Cb and Cr alternate in one plane, another workgroup writes the other component of every dword): the
// pairs of two positions in one 8-byte access, this component's halves out of them; stores go sample by sample
__device__ inline uint32_t ld_coh_c2(const int16_t* pair, int half) {              // pair: 8-byte aligned, the (Cb, Cr) of two positions
  const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pair), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_amdgcn_perm((uint32_t)(w >> 32), (uint32_t)w, half ? 0x07060302u : 0x05040100u);
}
__device__ inline void st_coh_c2(int16_t* p, uint32_t v) {                         // two samples of one component, kCStep apart
#if defined(INTRA_EXP) && (INTRA_EXP & 32)
  *(uint16_t*)p = (uint16_t)(v & 0xffffu); *(uint16_t*)(p + kCStep) = (uint16_t)(v >> 16);
  return;
#endif
  __hip_atomic_store(reinterpret_cast<uint16_t*>(p), (uint16_t)(v & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<uint16_t*>(p + kCStep), (uint16_t)(v >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline int ld_sample(const int16_t* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t w = ld_coh(reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3));
  return (int)(int16_t)((a & 2) ? (w >> 16) : (w & 0xffffu));
}

#ifndef INTRA_LEAN_OCC
#define INTRA_LEAN_OCC 4                 // waves per SIMD the LEAN kernel is compiled for (128 VGPRs; it would take 130)
#endif
#ifndef INTRA_MANY_OCC
#define INTRA_MANY_OCC 4                 // ... and the three-wave kernel of batches (128 VGPRs and 44 bytes of scratch instead of 147: a fifth workgroup per CU,
#endif                                   // sixteen I pictures 7.38 -> 6.82 ms; the eight-wave kernel of single pictures is latency, not residency: left alone)
#ifndef INTRA_SPARSE_MAX
#define INTRA_SPARSE_MAX 16              // in 64ths of a CTU's 8x8 areas: at most this share intra -> the CTU is not staged (k_intra)
#endif

struct IntraScratch {                    // what ONE TU in flight needs: one per wave
  int line[4 * 32 + 4];                  // reference line: [0,2N) left column bottom-up, [2N] corner, (2N,4N] row above
  int filt[4 * 32 + 4];                  // the same after smoothing
  int proj[3 * 32 + 4];                  // angular modes: main reference incl. the projected side samples, index k + 32
#ifdef INTRA_TIMING2                 // diagnostic build: shader-clock cycles per phase of the TU path, summed per wave (printed by k_intra)
  unsigned tm[16];
#endif
};
// LEAN: the kernel for calls whose CTUs are ALL taken as sparse (k_intra): no copy of the samples and of the residual at all
#ifdef INTRA_TIMING2
#define TK_START() long long tk0 = clock64()
#define TK(k) { const long long tkc = clock64(); if ((threadIdx.x & 63) == 0) W.tm[k] += (unsigned)(tkc - tk0); tk0 = tkc; }
#else
#define TK_START()
#define TK(k)
#endif

template <bool LEAN>
struct IntraLdsT {
  static constexpr bool lean = LEAN;
  // the CTU's samples of this component (columns -2..63 at index x + 2) and the row above it (columns -2..127): references
  // inside the CTU never leave the chip, and a TU does not wait for its stores before the next one starts
  __attribute__((aligned(4))) int16_t pix[LEAN ? 1 : 64][66];
  __attribute__((aligned(4))) int16_t top[LEAN ? 2 : 132];
  // the CTU's TComDataCU arrays, fetched once with one dword per lane and array (the walk over CUs and TUs is a serial chain:
  // every byte it had to wait for from global memory would cost a round trip)
  // the CTU's residual of this component -- de-quantisation and inverse transform do not depend on the neighbours: k_itx has done them
  // for every coded TU before this kernel starts --, as PicDev::resid lays it out (8x8 tiles of 128 bytes, rows in resid_slot order,
  // tile (tx, ty) of the CTU at index ty * tiles per CTU row + tx), staged while the block still waits for its neighbours
  __attribute__((aligned(16))) int16_t res[LEAN ? 8 : 64 * 64];
  __attribute__((aligned(4))) uint8_t m_depth[256], m_part[256], m_pred[256], m_tr[256], m_cbf[256], m_dir[256], m_dirl[256], m_byp[256], m_pcm[256];
};

struct TuCtx {
  int comp, ctu, z_tu, log2n, mode, cbf, bypass, x0, y0;          // x0, y0: component samples
  int cip, slice, tile, nb_same;
  // picture constants, read from the descriptor ONCE per CTU (through `P` every use is a fresh scalar load: stores may alias the descriptor)
  int16_t* plane; int pitch, bd, log2ctu, rext, strong;
  int cs, fmt;                                                    // chroma at half size (4:2:0 chroma); chroma_format_idc
  int sparse;                                                     // the CTU has no LDS copy of its samples and residual (k_intra): both straight from the picture
  const int16_t* resid; int rtw;                                  // this component's residual tiles, tiles per picture row
  unsigned long long am;                                          // availability of the TU's 4U + 1 reference units (IntraSched::avail)
  int sub_lo, sub_hi;                                             // available units form ONE run: substitution = clamping the line index to [sub_lo, sub_hi]; else sub_lo < 0                             // per-CTU constants: constrained intra pred, chroma QP offset, slice / tile index, neighbours in the same slice and tile
  int cx0, cy0;                                                   // CTU origin in component samples
};

// availability of the 4x4 luma partition at luma sample (px, py) as intra reference of the TU at (ctu, z_tu)
// nb_same: bit k set = neighbouring CTU k (0 left, 1 above-left, 2 above, 3 above-right) lies in the same slice and tile, worked out
// once per CTU; m_pred: the CTU's own prediction modes in LDS (constrained intra prediction)
__device__ __attribute__((always_inline)) inline bool intra_avail(const PicDev& P, int ctu, int z_tu, int px, int py, bool cip, unsigned nb_same, const uint8_t* m_pred) {
  if (px < 0 || py < 0 || px >= P.width || py >= P.height) return false;
  const int ctu_mask = (1 << P.log2ctu) - 1;
  const int nctu = (py >> P.log2ctu) * P.ctus_w + (px >> P.log2ctu);
  const int bx = (px & ctu_mask) >> 2, by = (py & ctu_mask) >> 2;
  int nz = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) nz |= (((bx >> k) & 1) << (2 * k)) | (((by >> k) & 1) << (2 * k + 1));
  if (nctu == ctu) {
    if (nz >= z_tu) return false;
    return !cip || m_pred[nz] == HMGPU_MODE_INTRA;
  }
  if (nctu > ctu) return false;
  const int k = nctu == ctu - 1 ? 0 : nctu == ctu - P.ctus_w - 1 ? 1 : nctu == ctu - P.ctus_w ? 2 : 3;
  if (!((nb_same >> k) & 1)) return false;
  if (cip && ldg(P.pred_mode + (size_t)nctu * P.parts + nz) != HMGPU_MODE_INTRA) return false;
  return true;
}

// sum over the first N lanes (the others hold 0), in every one of them: log2 N exchanges, not six -- each is a trip through the LDS crossbar
template <int N>
__device__ inline int wave_sum(int v) {
#pragma unroll
  for (int o = N / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one TU: everything between "the neighbours are reconstructed" and "this TU is reconstructed"
template <int LOG2N, class IntraLds>
__device__ __attribute__((always_inline)) inline void intra_tu(const PicDev& P, const TuCtx& t, IntraLds& L, IntraScratch& W) {
  const bool sparse = IntraLds::lean || t.sparse;
  constexpr int N = 1 << LOG2N;
  const int lane = threadIdx.x & 63;
  const int comp = t.comp, cs = t.cs;                       // (4:2:0: chroma at half size; 4:4:4: like luma.  4:2:2 does not come here)
  const int bd = t.bd, maxv = (1 << bd) - 1;
  const int pitch = t.pitch;
  int16_t* plane = t.plane;
  const int us = 4 >> cs, lus = 2 - cs, U = N >> lus;      // samples per availability unit (and its log2), units per TU side
  const int corner = 2 * N, total = 4 * N + 1;

  const int n = lane & (N - 1);
  const bool active = lane < N;
  TK_START();

  // the residual of row n: from the CTU's staged tiles (D.); a sparse CTU has none: from the picture's tiles, on its way while the
  // reference line is built (k_itx computed it in an earlier launch; RExt rotation / RDPCM included)
  uint32_t res[N / 2];
#pragma unroll
  for (int i = 0; i < N / 2; i++) res[i] = 0;
  if (sparse && t.cbf && active) {
    const int ry = t.y0 + n;
    const int16_t* r = t.resid + ((size_t)(ry >> 3) * t.rtw + (t.x0 >> 3)) * 64 + resid_slot(ry) * 8;
    if constexpr (N == 4) { const u32x2 a = ldg2(r + (t.x0 & 4)); res[0] = a.x; res[1] = a.y; }
    else {
#pragma unroll
      for (int i = 0; i < N / 8; i++) { const u32x4 a = ldg4(r + i * 64); res[4 * i] = a.x; res[4 * i + 1] = a.y; res[4 * i + 2] = a.z; res[4 * i + 3] = a.w; }
    }
  }

  // ---- A. the reference line.  Availability per unit -- units [0, 2U) left column bottom-up, 2U the corner, (2U, 4U] the row above -- is a
  // property of the TU's position: worked out for the whole list when the CTU starts (IntraSched::avail), not on the serial chain
  const unsigned long long am = t.am;
  // the sample line[i] takes: its position in the plane (component samples); false: nothing is available (the default value)
  auto ref_src = [&](int i, int& sx, int& sy) {
    if (!am) return false;
    const int u = i < corner ? i >> lus : (i == corner ? 2 * U : 2 * U + 1 + ((i - corner - 1) >> lus));
    int src = i;
    if (t.sub_lo >= 0) src = min(max(i, t.sub_lo), t.sub_hi);      // one run of available units (all available included): padding is a clamp
    else if (!((am >> u) & 1)) {
      const unsigned long long lower = am & ((1ull << u) - 1);
      int j, last;
      if (lower) { j = 63 - __builtin_clzll(lower); last = 1; } else { j = __builtin_ctzll(am); last = 0; }
      const int first_of = j < 2 * U ? j << lus : (j == 2 * U ? corner : corner + 1 + ((j - 2 * U - 1) << lus));
      src = first_of + ((last && j != 2 * U) ? us - 1 : 0);
    }
    // (left column bottom-up, the corner, the row above -- by selects, not by three branches)
    sx = src <= corner ? t.x0 - 1 : t.x0 + (src - corner - 1);
    sy = src < corner ? t.y0 + (corner - 1 - src) : t.y0 - 1;
    return true;
  };
  if (sparse) {
    // no copy of the CTU in LDS: every sample with a coherent load from the picture, all of a lane's loads in flight together
    constexpr int K = (4 * N + 1 + 63) / 64;
    int v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = lane + 64 * k;
      int sx, sy;
      v[k] = 1 << (bd - 1);
      if (i < total && ref_src(i, sx, sy)) v[k] = ld_sample(plane + (ptrdiff_t)sy * pitch + (comp ? kCStep : 1) * sx);
    }
#pragma unroll
    for (int k = 0; k < K; k++) if (lane + 64 * k < total) W.line[lane + 64 * k] = v[k];
  } else
  for (int i = lane; i < total; i += 64) {
    int v = 1 << (bd - 1);
    int sx, sy;
    if (ref_src(i, sx, sy)) {
      // position of the source sample relative to the CTU: row -1 lives in top[], everything else in pix[]
      sx -= t.cx0; sy -= t.cy0;
      const int16_t* at = sy < 0 ? &L.top[sx + 2] : &L.pix[sy][sx + 2];       // (one load, its address selected)
      v = *at;
    }
    W.line[i] = v;
  }
  wave_lds_sync();
  TK(0)

  // ---- B. smoothing (filteringIntraReferenceSamples + initAdiPatternChType); most TUs are not smoothed (chroma, 4x4, DC,
  // the modes near horizontal / vertical) and predict straight from line[]: one LDS round trip less on the serial chain
  const int thr = LOG2N == 2 ? 10 : LOG2N == 3 ? 7 : LOG2N == 4 ? 1 : 0;
  // (filterIntraReferenceSamples, TComChromaFormat.h:150-153: luma, and chroma where it is not subsampled)
  const bool filt = (comp == 0 || t.fmt == 3) && t.mode != 1 && min(abs(t.mode - 10), abs(t.mode - 26)) > thr && !(t.rext & HMGPU_REXT_INTRA_SMOOTHING_DISABLED);
#if defined(INTRA_EXP) && (INTRA_EXP & 2)      // experiment: no smoothing pass
  if (false) {
#else
  if (filt) {
#endif
    bool strong = false;
    int bl = 0, tl = 0, tr = 0;
    if (N == 32 && t.strong && comp == 0) {                 // (strong smoothing: luma only, TComPattern.cpp:196)
      bl = W.line[0]; tl = W.line[corner]; tr = W.line[total - 1];
      const int th = 1 << (bd - 5);
      strong = abs(bl + tl - 2 * W.line[N]) < th && abs(tl + tr - 2 * W.line[corner + N]) < th;
    }
    for (int i = lane; i < total; i += 64) {
      // (the three samples always, the ends of the line with themselves as neighbours and kept as they are: loads that hang on no
      // condition leave together)
      const int lo = W.line[max(i - 1, 0)], mid = W.line[i], hi = W.line[min(i + 1, total - 1)];
      int v = mid;
      if (i > 0 && i < total - 1) {
        if (strong) {
          // (24-bit multiplies throughout the prediction: samples and weights are small, v_mul_lo_u32 runs at quarter rate)
          if (i < corner) v = (__mul24(2 * N - i, bl) + __mul24(i, tl) + N) >> (LOG2N + 1);
          else if (i > corner) v = (__mul24(2 * N - (i - corner), tl) + __mul24(i - corner, tr) + N) >> (LOG2N + 1);
        } else {
          v = (lo + 2 * mid + hi + 2) >> 2;
        }
      }
      W.filt[i] = v;
    }
    wave_lds_sync();
  }
  TK(1)

  // ---- C. prediction of row n by lane n
#if defined(INTRA_EXP) && (INTRA_EXP & 2)
  const int* f = W.line;
#else
  const int* f = filt ? W.filt : W.line;
#endif

  const bool edge = comp == 0 && N <= 16;                   // MAXIMUM_INTRA_FILTERED_WIDTH (TypeDef.h:117)
  // implicit RDPCM in a lossless CU: horizontal / vertical prediction without its edge filter (TComPrediction.cpp:476)
  const bool edge_ang = edge && !(t.bypass && (t.rext & HMGPU_REXT_IMPLICIT_RDPCM));
  int p[N];
#if defined(INTRA_EXP) && (INTRA_EXP & 8)      // experiment (wrong samples): the row above copied down -- what the mode-specific prediction code costs
  if (true) {
#pragma unroll
    for (int x = 0; x < N; x++) p[x] = f[corner + 1 + x];
  } else
#endif
  if (t.mode == 0) {
    const int left = f[corner - 1 - n], bl = f[corner - 1 - N], tr = f[corner + 1 + N];
#pragma unroll
    for (int x = 0; x < N; x++) {
      const int ab = f[corner + 1 + x];
      p[x] = ((left << LOG2N) + N + __mul24(x + 1, tr - left) + (ab << LOG2N) + __mul24(n + 1, bl - ab)) >> (LOG2N + 1);
    }
  } else if (t.mode == 1) {
    const int dc = (wave_sum<N>(active ? f[corner + 1 + n] + f[corner - 1 - n] : 0) + N) >> (LOG2N + 1);
#pragma unroll
    for (int x = 0; x < N; x++) p[x] = dc;
    if (edge) {
      if (n == 0) {
#pragma unroll
        for (int x = 1; x < N; x++) p[x] = (f[corner + 1 + x] + 3 * dc + 2) >> 2;
        p[0] = (f[corner + 1] + f[corner - 1] + 2 * dc + 2) >> 2;
      } else {
        p[0] = (f[corner - 1 - n] + 3 * dc + 2) >> 2;
      }
    }
  } else {
    const bool ver = t.mode >= 18;
    const int am_ = ver ? t.mode - 26 : -(t.mode - 10);
    const int aa = abs(am_);
    // intraPredAngle {0, 2, 5, 9, 13, 17, 21, 26, 32} and its inverse {-, 4096, 1638, 910, 630, 482, 390, 315, 256} (TComPrediction.cpp:276-277) out of
    // packed constants: the chains of conditionals the tables were written as before compiled into ~20 scalar branches on the TU chain
    const int ang_abs = aa >= 8 ? 32 : (int)((0x1a15110d09050200ull >> (8 * aa)) & 0xffu);
    const int inv = aa >= 8 ? 256 : (int)(((aa & 4) ? 0x013b018601e20276ull : 0x038e066610000000ull) >> (16 * (aa & 3)) & 0xffffu);
    const int ang = am_ < 0 ? -ang_abs : ang_abs;
    const int sgn = ver ? 1 : -1;                           // MAIN(i) = f[corner + sgn*i], SIDE(i) = f[corner - sgn*i]
    if (ang == 0) {
      // pure vertical (26) / horizontal (10): a copy of the row above / the left column, with the edge filter on the first column / row of luma
      // TUs up to 16x16 (xPredIntraAng, TComPrediction.cpp:279-300) -- kept apart from the interpolating modes: a condition inside their loop
      // would tie every sample's loads to its own basic block, one trip to LDS per sample instead of one per row
      if (ver) {
#pragma unroll
        for (int x = 0; x < N; x++) p[x] = f[corner + 1 + x];
        if (edge_ang) p[0] = clip3(0, maxv, p[0] + ((f[corner - (n + 1)] - f[corner]) >> 1));
      } else {
        const int left = f[corner - 1 - n];
#pragma unroll
        for (int x = 0; x < N; x++) p[x] = left;
        if (edge_ang && n == 0) {
          const int c0 = f[corner];
#pragma unroll
          for (int x = 0; x < N; x++) p[x] = clip3(0, maxv, left + ((f[corner + 1 + x] - c0) >> 1));
        }
      }
    } else if (ang > 0) {
      // no projected side samples: the main reference is f[] itself (another LDS round trip less).  Both samples always: with df = 0 the
      // weights are 32 and 0, and loads that do not hang on a condition go out together (line[] / filt[] have room for the entry past 4N)
      if (ver) {
        const int pos = __mul24(n + 1, ang), di = pos >> 5, df = pos & 31;      // a lane's row: one offset, one weight
        int r[N + 1];
#pragma unroll
        for (int x = 0; x <= N; x++) r[x] = f[corner + 1 + di + x];
#pragma unroll
        for (int x = 0; x < N; x++) p[x] = (__mul24(32 - df, r[x]) + __mul24(df, r[x + 1]) + 16) >> 5;
      } else {
#pragma unroll
        for (int x = 0; x < N; x++) {
          const int pos = (x + 1) * ang, di = pos >> 5, df = pos & 31;           // the same for all lanes
          const int i0 = corner - (n + di + 1);
          p[x] = (__mul24(32 - df, f[i0]) + __mul24(df, f[i0 - 1]) + 16) >> 5;
        }
      }
    } else {
      // main reference with its extension: proj[k + 32], k in [-N, N]; main = row above for vertical modes, left column otherwise
      for (int k = lane - N; k <= N; k += 64) {
        // (one load whatever the side: the index chosen by arithmetic, not by a branch around two loads)
        const int off = k >= 0 ? k : -((128 + __mul24(-k, inv)) >> 8);
        const int v = f[corner + sgn * off];
        W.proj[k + 32] = (k >= 0 || k > ((N * ang) >> 5)) ? v : 0;
      }
      wave_lds_sync();
      const int* r = W.proj + 32;
      if (ver) {
        const int pos = __mul24(n + 1, ang), di = pos >> 5, df = pos & 31;      // a lane's row: one offset, one weight
        int q[N + 1];
#pragma unroll
        for (int x = 0; x <= N; x++) q[x] = r[x + di + 1];
#pragma unroll
        for (int x = 0; x < N; x++) p[x] = (__mul24(32 - df, q[x]) + __mul24(df, q[x + 1]) + 16) >> 5;
      } else {
#pragma unroll
        for (int x = 0; x < N; x++) {
          const int pos = (x + 1) * ang, di = pos >> 5, df = pos & 31;           // the same for all lanes
          p[x] = (__mul24(32 - df, r[n + di + 1]) + __mul24(df, r[n + di + 2]) + 16) >> 5;
        }
      }
    }
  }

  TK(2)
  // ---- D. residual of row n: from the CTU's staged tiles
#if defined(INTRA_EXP) && (INTRA_EXP & 1)      // experiment: no residual (wrong samples)
  if (false) {
#else
  if (!sparse && t.cbf && active) {
#endif
    const int rx = t.x0 - t.cx0, ry = t.y0 - t.cy0 + n;                    // inside the CTU, component samples
    const int tpr = ((1 << t.log2ctu) >> cs) >> 3;                          // tiles per CTU row
    const int16_t* r = &L.res[((ry >> 3) * tpr + (rx >> 3)) * 64 + resid_slot(ry) * 8];
    if constexpr (N == 4) { const u32x2 a = *reinterpret_cast<const u32x2*>(r + (rx & 4)); res[0] = a.x; res[1] = a.y; }
    else {
#pragma unroll
      for (int i = 0; i < N / 8; i++) { const u32x4 a = *reinterpret_cast<const u32x4*>(r + i * 64); res[4 * i] = a.x; res[4 * i + 1] = a.y; res[4 * i + 2] = a.z; res[4 * i + 3] = a.w; }
    }
  }

  TK(3)
  // ---- E. reconstruction of row n: into the LDS copy (what later TUs of this CTU predict from) and, two samples per
  // coherent dword store, into the picture (what other CTUs and the loop filters read; nobody here waits for it)
  if (active) {
    int16_t* prow = plane + (ptrdiff_t)(t.y0 + n) * pitch + (comp ? kCStep : 1) * t.x0;
    uint32_t* row = reinterpret_cast<uint32_t*>(prow);
    uint32_t* lrow = reinterpret_cast<uint32_t*>(&L.pix[t.y0 - t.cy0 + n][t.x0 - t.cx0 + 2]);
    const uint32_t maxv2 = (uint32_t)maxv * 0x10001u;
    // (a TU's rows start at multiples of four samples of planes whose pitch is a multiple of 64: 8-byte stores; chroma: sample by sample, the
    // other component's workgroup owns the other half of every dword)
    uint32_t v[N / 2];
#pragma unroll
    for (int x = 0; x < N; x += 2) v[x / 2] = pk_clip_u(pk_add_sat(cvt_pk_sat(p[x], p[x + 1]), res[x / 2]), maxv2);
    // (the three uniform conditions once per row, not once per four samples)
    if (!sparse) {
#pragma unroll
      for (int x = 0; x < N / 2; x++) lrow[x] = v[x];
    }
    if (comp == 0) {
#pragma unroll
      for (int x = 0; x < N / 2; x += 2) st_coh2(row + x, v[x], v[x + 1]);
    } else {
#pragma unroll
      for (int x = 0; x < N / 2; x++) st_coh_c2(prow + kCStep * 2 * x, v[x]);
    }
  }
  // (sparse CTU: whoever predicts from this TU next -- this wave or another -- loads its samples from the picture: acknowledged first)
  if (sparse) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_lds_sync();
  TK(4)
}

template <class IntraLds>
__device__ __attribute__((always_inline)) inline void intra_tu_any(const PicDev& P, const TuCtx& t, IntraLds& L, IntraScratch& W) {
  switch (t.log2n) {
    case 2: intra_tu<2>(P, t, L, W); break;
    case 3: intra_tu<3>(P, t, L, W); break;
    case 4: intra_tu<4>(P, t, L, W); break;
    default: intra_tu<5>(P, t, L, W); break;
  }
}

// what a CTU can fetch before its neighbours are finished: its TComDataCU arrays and its own samples (inter CUs are final --
// written by earlier launches --, intra ones get overwritten below)
template <class IntraLds>
__device__ __attribute__((always_inline)) inline void intra_stage(const PicDev& P, int comp, int ctu, IntraLds& L, bool pix) {
  const int parts = P.parts;
  const size_t base = (size_t)ctu * parts;
  const int ctu_x = (ctu % P.ctus_w) << P.log2ctu, ctu_y = (ctu / P.ctus_w) << P.log2ctu;
  const int cs = (comp && P.csx) ? 1 : 0;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64 && 4 * lane < parts) {
    const size_t o = base + 4 * lane;
    auto dw = [&](const void* p) { return ldg(reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(p) + o)); };
    const uint32_t a0 = dw(P.depth), a1 = dw(P.part_size), a2 = dw(P.pred_mode), a3 = dw(P.tr_idx), a5 = dw(P.cbf[comp]),
                   a7 = dw(P.intra_dir[comp ? 1 : 0]), a8 = dw(P.intra_dir[0]), a9 = dw(P.bypass), a10 = dw(P.ipcm);
    auto put = [&](uint8_t* d, uint32_t v) { *reinterpret_cast<uint32_t*>(d + 4 * lane) = v; };
    put(L.m_depth, a0); put(L.m_part, a1); put(L.m_pred, a2); put(L.m_tr, a3); put(L.m_cbf, a5);
    put(L.m_dir, a7); put(L.m_dirl, a8); put(L.m_byp, a9); put(L.m_pcm, a10);
  }
  if (pix) {
    const int S = (1 << P.log2ctu) >> cs;                   // CTU size in samples of this component
    // the interior was written by the MC / residual kernels (earlier launches): plain 16-byte loads
    if (comp == 0) {
      const int16_t* org = P.rec[0] + (ptrdiff_t)ctu_y * P.pitch[0] + ctu_x;
      const int vpr = S / 8;                                  // 16-byte vectors per row
      for (int i = threadIdx.x; i < S * vpr; i += blockDim.x) {
        const int r = i / vpr, v = i % vpr;
        const u32x4 q = ldg4(org + (ptrdiff_t)r * P.pitch[0] + 8 * v);
        uint32_t* d = reinterpret_cast<uint32_t*>(&L.pix[r][2 + 8 * v]);
        d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
      }
    } else {
      // a vector = the (Cb, Cr) of four positions: this component's halves
      const int16_t* org = P.rec[1] + (ptrdiff_t)(ctu_y >> cs) * P.pitch[1] + kCStep * (ctu_x >> cs);
      const uint32_t sel = comp == 2 ? 0x07060302u : 0x05040100u;
      const int vpr = S / 4;
      for (int i = threadIdx.x; i < S * vpr; i += blockDim.x) {
        const int r = i / vpr, v = i % vpr;
        const u32x4 q = ldg4(org + (ptrdiff_t)r * P.pitch[1] + 8 * v);
        uint32_t* d = reinterpret_cast<uint32_t*>(&L.pix[r][2 + 4 * v]);
        d[0] = __builtin_amdgcn_perm(q.y, q.x, sel); d[1] = __builtin_amdgcn_perm(q.w, q.z, sel);
      }
    }
  }
}

// ---- order of the TUs of a CTU, progress between CTUs --------------------------------------------------------------------------------
// Intra prediction fixes WHAT a TU reads (the reconstructed samples of the neighbouring 4x4 units that precede it in z order), not WHEN
// the TUs run: any order in which every TU comes after the units it reads gives HM's samples.  z order is the order that makes a
// CTU's right column and bottom row -- what its neighbours wait for -- come last (16x16 areas 5, 7, 13, 15 of 16).  So a wave
// lists the intra TUs of its CTU, works out for each the units left of and above it that it depends on, and runs, among the TUs whose
// units are final, the one highest up (raster priority: the picture is wider than high, and the right column of a row of TUs is final two
// rows of units later).  Finished units are kept as bit masks per row and per column of the CTU's 4x4-unit grid; the neighbouring
// CTUs' border units come from one published word per CTU and component:
//   PicDev::intra_done[comp][ctu] = units of the CTU's LAST COLUMN that are final (bit y) | units of its LAST ROW << 16 (bit 16 + x).
// A CTU no longer waits for its neighbours as a whole, and not in their z order either.
// kind 0: transform unit, 1: PCM coding unit (log2n: its size), 2: the first of the four 4x4 luma TUs of an 8x8 area -- it stands for all
// four in the list: one wave runs them one after the other (they are a dependent chain anyway) without going back to the list in between --,
// 3: the other three (list entries for their availability masks only, never pending)
struct TuRun { uint8_t z, z_cu, log2n, kind; };
// PACKED: the per-entry words squeezed (the LEAN kernel: LDS decides how many of its workgroups a CU holds); the general kernel keeps them
// plain -- unpacking them in the ready-scan cost an I picture 3 %
template <bool PACKED>
struct IntraSchedT {
  TuRun tu[256];
  // per list entry; packed (round 4: 4.5 instead of 7 KB -- the LEAN kernel's workgroups are 9.5 KB, 16 of them per CU) as follows:
  //   col    bits 0-15: bit y = unit (x4 - 1, y) of the column left of the TU must be final (the left CTU's last column when x4 = 0);
  //          bit 16 / 17: bit 32 of the row mask / of the availability mask
  //   row    bit c + 1: unit (c, y4 - 1), c = -1 .. 31, of the row above the TU (the row of the CTUs above when y4 = 0), bits 0-31
  //   avl    bit u: reference unit u of the TU (intra_tu's numbering) is available (6.4.1 + constrained intra prediction), bits 0-31
  //   clamp  the run of available units as line indices (at most 128): first sample | last sample << 8 (0xffff: not one run)
  // (plain: col = the 16 bits, row / avl 64 bits, clamp = first | last << 16, 0xffffffff: not one run)
  using Wide = std::conditional_t<PACKED, uint32_t, uint64_t>;
  using Clamp = std::conditional_t<PACKED, uint16_t, uint32_t>;
  uint32_t e_col[256];
  Wide e_row[256], e_avl[256];
  Clamp e_clamp[256];
  // bits 24-31 of col (both forms): what the TU chain wants to know about the TU besides its place -- prediction mode (6 bits, DM_CHROMA resolved),
  // coded flag, lossless flag -- looked up once, when the list is built, instead of byte by byte out of the CTU's arrays on the chain
  __device__ void put(int i, uint32_t nc, uint64_t nr, uint64_t am, uint32_t cl, uint32_t desc) {
    if constexpr (PACKED) {
      e_col[i] = (nc & 0xffffu) | ((uint32_t)(nr >> 32) & 1u) << 16 | ((uint32_t)(am >> 32) & 1u) << 17 | desc << 24;
      e_row[i] = (uint32_t)nr; e_avl[i] = (uint32_t)am;
      e_clamp[i] = cl == 0xffffffffu ? (uint16_t)0xffffu : (uint16_t)((cl & 0xffu) | ((cl >> 16) << 8));
    } else { e_col[i] = (nc & 0xffffu) | desc << 24; e_row[i] = nr; e_avl[i] = am; e_clamp[i] = cl; }
  }
  __device__ uint32_t need_col(int i) const { return e_col[i] & 0xffffu; }
  __device__ uint64_t need_row(int i) const { if constexpr (PACKED) return (uint64_t)e_row[i] | ((uint64_t)((e_col[i] >> 16) & 1u) << 32); else return e_row[i]; }
  __device__ uint64_t avail(int i) const { if constexpr (PACKED) return (uint64_t)e_avl[i] | ((uint64_t)((e_col[i] >> 17) & 1u) << 32); else return e_avl[i]; }
  __device__ uint32_t clampi(int i) const {
    if constexpr (PACKED) { const uint32_t v = e_clamp[i]; return v == 0xffffu ? 0xffffffffu : (v & 0xffu) | ((v >> 8) << 16); } else return e_clamp[i];
  }
  uint32_t done_r[16], done_c[16];   // final units per row (bit x) / per column (bit y); set with LDS atomics by the wave that finished a TU
  uint32_t got[3];                   // border units of the neighbouring CTUs whose samples are in LDS (L.pix columns 0, 1 / L.top): column | row lo | row hi
  uint32_t pend[8];                  // list entries nobody has taken yet (bit i & 31 of pend[i >> 5]); a wave takes one with an atomic AND
  uint8_t member[64][4];             // list entries of the four 4x4 luma TUs of every 8x8 area (kinds 2 and 3), by z >> 2 and z & 3
  int32_t n_tus;                     // list length
#ifdef INTRA_TIMING                  // diagnostic build: where the time of a CTU goes (printed by k_intra for one CTU row)
  unsigned long long t_tu, t_claim, t_post, t_idle, t_first, t_last, t_a, t_b, t_c;
  uint32_t n_run, n_idle;
#endif
};

struct Neighbours {
  uint32_t* prog;            // intra_done of this component
  int ctu[4];                // left, above-left, above, above-right (-1: none)
  bool wait[4];              // the neighbour is reconstructed by this launch and its border towards this CTU holds intra samples
  uint32_t ext_col;          // final units of the left CTU's last column
  uint64_t ext_row;          // final units of the row above the CTU: bit 0 above-left CTU's corner unit, 1..16 above, 17..32 above-right
  bool broken;               // a wait gave up (fault flagged): no further waiting in this block
};

__device__ inline int z_of(int x, int y) {
  int z = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) z |= (((x >> k) & 1) << (2 * k)) | (((y >> k) & 1) << (2 * k + 1));
  return z;
}

// Every sample this kernel stores is an agent-scope atomic store (written through to the level all XCDs read from) and every
// neighbour sample it loads an agent-scope atomic load, so no cache has to be written back or invalidated around the progress word: the
// producer waits until its stores are acknowledged, the consumer issues its loads after it has seen the bits.  (Agent-scope
// release / acquire FENCES also write back / invalidate L2 for ordinary accesses: per publication that cost more than it gained.)
__device__ __attribute__((always_inline)) inline void publish_progress(uint32_t* prog, int ctu, uint32_t word) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(prog + ctu, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the neighbours' published words -> ext_col / ext_row
__device__ __attribute__((always_inline)) inline void poll_neighbours(const PicDev& P, Neighbours& nb, int pw) {
  const uint32_t full = (1u << pw) - 1u;
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = nb.wait[k] ? __hip_atomic_load(nb.prog + nb.ctu[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
  asm volatile("" ::: "memory");
  nb.ext_col = w[0] & full;
  nb.ext_row = (uint64_t)((w[1] >> (16 + pw - 1)) & 1u) | ((uint64_t)((w[2] >> 16) & full) << 1) | ((uint64_t)((w[3] >> 16) & full) << (1 + pw));
}

// samples of the neighbouring CTUs' border units the TU needs (all final: the TU was ready) into LDS, unless there already
// (the masks of what has been fetched are shared by the waves of the workgroup: a wave with nothing to run fetches what has become final
// next door, so that the TUs along the CTU's left and top border find their reference samples in LDS)
// org: the CTU's first sample of this component in the picture, pitch: the plane's
template <class IntraLds, class IntraSched>
__device__ __attribute__((always_inline)) inline void fetch_border(const int16_t* org, int pitch, int comp, int cs, uint32_t need_col, uint64_t need_row, IntraSched& Q, IntraLds& L) {
  const int lane = threadIdx.x & 63, us = 4 >> cs;
  const uint32_t mc = need_col & ~Q.got[0];
  const uint64_t mr = need_row & ~((uint64_t)Q.got[1] | ((uint64_t)Q.got[2] << 32));
  if (!mc && !mr) return;
  if (mc) {
    // unit u = rows u * us .. + us - 1 of the two columns left of the CTU: one dword per row
    const int u = lane >> 2, r = lane & 3;
    if (((mc >> u) & 1) && r < us) {
      const int row = u * us + r;
      // (chroma: org = the CTU's first (Cb, Cr) pair, positions -2, -1 are the eight bytes in front of the row)
      reinterpret_cast<uint32_t*>(&L.pix[row][0])[0] = comp == 0 ? ld_coh(reinterpret_cast<const uint32_t*>(org + (ptrdiff_t)row * pitch - 2))
                                                                 : ld_coh_c2(org + (ptrdiff_t)row * pitch - 2 * kCStep, comp - 1);
    }
  }
  if (mr) {
    // dword d of top[] = columns 2d - 2, 2d - 1; unit c (-1 .. 31) = columns c * us .. + us - 1
    const int dwords = 1 + (33 * us) / 2;
    for (int d = lane; d < dwords && d < 66; d += 64) {
      const int col = 2 * d - 2;                            // first column of the dword
      const int c = col < 0 ? -1 : col >> (2 - cs);
      if ((mr >> (c + 1)) & 1) reinterpret_cast<uint32_t*>(L.top)[d] = comp == 0 ? ld_coh(reinterpret_cast<const uint32_t*>(org - pitch - 2) + d)
                                                                                    : ld_coh_c2(org - pitch + 2 * kCStep * (d - 1), comp - 1);
    }
  }
  wave_lds_sync();
  if (lane == 0) {
    if (mc) atomicOr(&Q.got[0], mc);
    if (mr) { atomicOr(&Q.got[1], (uint32_t)mr); atomicOr(&Q.got[2], (uint32_t)(mr >> 32)); }
  }
}

// all intra CUs of one CTU, one component (xReconIntraQT per CU, xIntraRecQT over its TU tree), in dependency order, by the
// waves of the workgroup: wave 0 lists the TUs and what they depend on, then every wave takes, again and again, the first TU of the list
// that is ready and that nobody has taken (an atomic AND on the pending mask), runs it and marks its units final (atomic ORs).  TUs that
// do not depend on each other -- the next ones along an anti-diagonal of the CTU -- run side by side; the CTU's samples, the done masks
// and the list are shared in LDS, the reference line / transform scratch of a TU in flight is the wave's own (IntraScratch).
template <class IntraLds, class IntraSched>
__device__ __attribute__((always_inline)) inline void intra_ctu(const PicDev& P, int comp, int ctu, bool sparse_ctu, IntraLds& L, IntraSched& Q, IntraScratch& W, Neighbours& nb) {
  const bool sparse = IntraLds::lean || sparse_ctu;
  const int parts = P.parts, pw = P.pw;
  const int ctu_x = (ctu % P.ctus_w) << P.log2ctu, ctu_y = (ctu / P.ctus_w) << P.log2ctu;
  const int cs = (comp && P.csx) ? 1 : 0;
  const int lane = threadIdx.x & 63;
  const int slice = ldg(P.slice_idx + ctu), tile = ldg(P.tile_idx + ctu);
  const SliceDev& sd = P.slices[slice];
  const int cip = ldg(&sd.constrained_intra_pred);
  // which of the four neighbouring CTUs an intra reference may come from (same slice, same tile): once per CTU
  int nb_same = 0;
  {
    const int cx = ctu % P.ctus_w, cy = ctu / P.ctus_w;
    const int nbc[4] = {cx > 0 ? ctu - 1 : -1, (cx > 0 && cy > 0) ? ctu - P.ctus_w - 1 : -1, cy > 0 ? ctu - P.ctus_w : -1,
                        (cy > 0 && cx + 1 < P.ctus_w) ? ctu - P.ctus_w + 1 : -1};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (nbc[k] >= 0 && ldg(P.slice_idx + nbc[k]) == slice && ldg(P.tile_idx + nbc[k]) == tile) nb_same |= 1 << k;
  }
  // the picture constants the TU chain needs, out of the descriptor once
  const int h_bd = P.bd[comp], h_pitch = P.pitch[comp], h_log2ctu = P.log2ctu, h_rext = P.range_ext, h_strong = P.strong_intra_smoothing, h_fmt = P.fmt;
  int16_t* const h_plane = P.rec[comp];
  const int16_t* const h_resid = P.resid[comp];
  const int h_rtw = (P.grid_w / 2) >> cs;
  // the CTU's first sample in the plane; chroma: its first (Cb, Cr) pair in the plane of both components (fetch_border reads whole pairs)
  const int16_t* const org = comp == 0 ? h_plane + (ptrdiff_t)ctu_y * h_pitch + ctu_x : P.rec[1] + (ptrdiff_t)(ctu_y >> cs) * h_pitch + kCStep * (ctu_x >> cs);
  const int wv = threadIdx.x >> 6;
  auto mark_done = [&](int x4, int y4, int U) {           // units [x4, x4 + U) x [y4, y4 + U) are final
    if (lane < 16) {
      if (lane >= y4 && lane < y4 + U) atomicOr(&Q.done_r[lane], ((1u << U) - 1u) << x4);
    } else if (lane < 32) {
      const int c = lane - 16;
      if (c >= x4 && c < x4 + U) atomicOr(&Q.done_c[c], ((1u << U) - 1u) << y4);
    }
  };
  auto footprint = [&](const TuRun& e, int& x4, int& y4, int& U) {     // in 4x4 luma units of the CTU
    x4 = zscan_x(e.z); y4 = zscan_y(e.z);
    U = e.kind ? (1 << (e.log2n - 2)) : max(1, ((1 << e.log2n) << cs) >> 2);
  };
#if defined(INTRA_STOP) && INTRA_STOP == 6
  if (IntraLds::lean && pw > 0 && nb_same >= 0 && h_plane != nullptr && h_resid != nullptr) return;
#endif
  // mode | coded << 6 | lossless << 7 of the TU of list entry e (IntraSched::put)
  auto tu_desc = [&](const TuRun& e) -> uint32_t {
    if (e.kind == 1) return 0u;
    const int zs = e.z, tr = L.m_tr[zs];
    int mode = L.m_dir[zs];
    // DM_CHROMA_IDX (TDecCu.cpp:523-524, getChromasCorrespondingPULumaIdx): the luma mode of the CU's first partition; 4:4:4: of the block's own partition
    if (comp && mode == 36) mode = L.m_dirl[h_fmt == 3 ? zs : e.z_cu];
    // (4:4:4: the tiles hold zeros where nothing is coded and the cross-component term where only that is: always added)
    const uint32_t cbf = (comp && h_fmt != 1) ? 1u : (uint32_t)((L.m_cbf[zs] >> tr) & 1);
    return (uint32_t)(mode & 63) | cbf << 6 | (L.m_byp[zs] ? 1u << 7 : 0u);
  };
  if (wv == 0) {
  // ---- 1. the list, in raster order of the TU origins (= the priority they run in), one lane per 8x8 AREA of the CTU (a coding unit is at
  // least that large: prediction mode, transform depth and PCM flag are the same in the area's four partitions).  An area starts a TU of
  // 8x8 or more when its z index is aligned to the TU's size, or holds the four 4x4 luma TUs of an 8x8 CU (kinds 2, 3, 3, 3 -- with
  // subsampled chroma their one 4x4 chroma TU).  Areas that are not reconstructed here (inter CUs, outside the picture) are final from the
  // start.  (Round 4: this walk was lane-parallel over the 256 4x4 units, four rounds of two passes -- 3-4 of a sparse CTU's ~15 us.)
  int n_tus = 0, n_sched = 0;
  {
    const int paw = pw >> 1, log2paw = P.log2ctu - 3, areas = paw * paw;       // areas per CTU row (8, 4, 2)
    const int pic_w = P.width, pic_h = P.height;
    const bool has_pcm = P.pcm[comp] != nullptr;
    bool preset[2] = {false, false};                          // raster pass (x fastest) and transposed pass (y fastest): final from the start
    TuRun e = {0, 0, 0, 0};
    int cnt = 0;                                              // list entries of this area: 0, 1 or 4
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int a0 = lane & (paw - 1), a1 = lane >> log2paw;
      const int ax = pass ? a1 : a0, ay = pass ? a0 : a1;
      if (lane >= areas) continue;
      const int z = z_of(2 * ax, 2 * ay);
      if (ctu_x + 8 * ax >= pic_w || ctu_y + 8 * ay >= pic_h || (int8_t)L.m_part[z] == HMGPU_SIZE_NONE || (int8_t)L.m_pred[z] != HMGPU_MODE_INTRA) { preset[pass] = true; continue; }
      if (pass) continue;
      const int depth = L.m_depth[z];
      const int cu_parts = parts >> (2 * depth), log2cu = P.log2ctu - depth;
      const int z_cu = z & ~(cu_parts - 1);
      if (L.m_pcm[z_cu] && has_pcm) {
        cnt = z == z_cu;
        e = TuRun{(uint8_t)z, (uint8_t)z_cu, (uint8_t)log2cu, 1};
      } else {
        const int log2tu = log2cu - L.m_tr[z];
        if (log2tu >= 3) {
          cnt = (z & ((1 << (2 * (log2tu - 2))) - 1)) == 0;
          e = TuRun{(uint8_t)z, (uint8_t)z_cu, (uint8_t)(log2tu - cs), 0};
        } else if (cs) {
          // subsampled chroma: the four 4x4 luma TUs share one 4x4 chroma TU, with the first of them (TComTU.cpp:141-171)
          cnt = 1;
          e = TuRun{(uint8_t)z, (uint8_t)z_cu, 2, 0};
        } else {
          cnt = 4;                                            // (4:4:4 chroma: the luma TUs' twins)
          e = TuRun{(uint8_t)z, (uint8_t)z_cu, 2, 2};
        }
      }
    }
#if defined(INTRA_STOP) && INTRA_STOP == 7
    if (IntraLds::lean && pw > 0) { if (cnt) Q.tu[lane] = e; return; }
#endif
    // the done masks per row / column of UNITS: an area's bit for both of its units
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(preset[0]), m1 = __builtin_amdgcn_ballot_w64(preset[1]);
    if (lane < pw) {
      auto twice = [](uint32_t x) { x = (x | (x << 4)) & 0x0f0fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u; return x | (x << 1); };
      const int sh = (lane >> 1) * paw;
      Q.done_r[lane] = twice((uint32_t)(m0 >> sh) & ((1u << paw) - 1u));
      Q.done_c[lane] = twice((uint32_t)(m1 >> sh) & ((1u << paw) - 1u));
    }
    // What waves can take -- TUs of 8x8 and more, PCM CUs, the heads of the 4x4 groups: at most one per area, 64 -- comes first, in raster order; the
    // groups' other three entries (never pending: list entries for their availability words only) behind them.  A scan is ONE round of 64 lanes whatever
    // the CTU holds.
    const unsigned long long c1 = __builtin_amdgcn_ballot_w64(cnt == 1), c4 = __builtin_amdgcn_ballot_w64(cnt == 4);
    const unsigned long long below = (1ull << lane) - 1ull;
    n_sched = __popcll(c1 | c4);
    if (cnt) Q.tu[__popcll((c1 | c4) & below)] = e;
    if (cnt == 4) {
      const int off = n_sched + 3 * __popcll(c4 & below);
#pragma unroll
      for (int j = 1; j < 4; j++) Q.tu[off + j - 1] = TuRun{(uint8_t)(e.z + j), e.z_cu, 2, 3};
    }
    n_tus = n_sched + 3 * __popcll(c4);
  }
  wave_lds_sync();
#if defined(INTRA_STOP) && INTRA_STOP == 4     // (LEAN kernel only: one wave, nobody is left at the barrier)
  if (IntraLds::lean && pw > 0) return;
#endif
  // ---- 2. what each TU depends on.  A short list (scattered intra CUs in a P picture, a CTU of a few large TUs): the TUs one after the other, the
  // lanes over the units of each -- a 32x32 TU has 33 reference units and as many column / row units to test, which one lane walking them
  // alone made the longest part of a sparse CTU's life (6-8 of ~18 us, INTRA_TIMING).  A long list: the lanes over the TUs, as before.
  if (n_tus <= 12) {
    for (int i = 0; i < n_tus; i++) {
      const TuRun e = __builtin_bit_cast(TuRun, (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, Q.tu[i])));
      int x4, y4, U;
      footprint(e, x4, y4, U);
      if (lane == 0 && e.kind >= 2) Q.member[e.z >> 2][e.z & 3] = (uint8_t)i;
      bool colbit = false, rowbit = false, ab = false;
      if (e.kind == 0 || e.kind == 2) {
        const int Un = e.kind == 2 ? 2 : U;
        const int y = lane, c = lane - 1;                      // lane y: unit (x4 - 1, y) of the column; lane c + 1: unit (c, y4 - 1) of the row
        colbit = y < pw && y >= (y4 > 0 ? y4 - 1 : 0) && y < y4 + 2 * Un && (x4 == 0 || z_of(x4 - 1, y) < e.z);
        rowbit = y4 == 0 ? (c >= x4 - 1 && c < x4 + 2 * Un) : (c >= x4 && c < min(pw, x4 + 2 * Un) && z_of(c, y4 - 1) < e.z);
      }
      if (e.kind != 1 && lane <= 4 * U) {
        const int lx = ctu_x + 4 * x4, ly = ctu_y + 4 * y4, u = lane;
        int px, py;
        if (u < 2 * U) { px = lx - 4; py = ly + 4 * (2 * U - 1 - u); }
        else if (u == 2 * U) { px = lx - 4; py = ly - 4; }
        else { px = lx + 4 * (u - 2 * U - 1); py = ly - 4; }
#if defined(INTRA_EXP) && (INTRA_EXP & 4)
        ab = px >= 0 && py >= 0;
#else
        ab = intra_avail(P, ctu, e.z, px, py, cip != 0, (unsigned)nb_same, L.m_pred);
#endif
      }
      const uint32_t nc = (uint32_t)__builtin_amdgcn_ballot_w64(colbit);
      const uint64_t nr = __builtin_amdgcn_ballot_w64(rowbit), am = __builtin_amdgcn_ballot_w64(ab);
      uint32_t cl = 0xffffffffu;
      if (am != 0) {
        const int j0 = __builtin_ctzll(am), j1 = 63 - __builtin_clzll(am);
        if ((am >> j0) == (2ull << (j1 - j0)) - 1ull) {
          const int us = 4 >> cs, nn = U * us, corner = 2 * nn;
          const int lo = j0 < 2 * U ? j0 * us : (j0 == 2 * U ? corner : corner + 1 + (j0 - 2 * U - 1) * us);
          const int hi = j1 < 2 * U ? j1 * us + us - 1 : (j1 == 2 * U ? corner : corner + 1 + (j1 - 2 * U - 1) * us + us - 1);
          cl = (uint32_t)lo | ((uint32_t)hi << 16);
        }
      }
      if (lane == 0) Q.put(i, nc, nr, am, cl, tu_desc(e));
    }
  } else
  for (int i = lane; i < n_tus; i += 64) {
    const TuRun e = Q.tu[i];
    int x4, y4, U;
    footprint(e, x4, y4, U);
    uint32_t nc = 0;
    uint64_t nr = 0;
    if (e.kind >= 2) Q.member[e.z >> 2][e.z & 3] = (uint8_t)i;
    if (e.kind == 0 || e.kind == 2) {
      const int Un = e.kind == 2 ? 2 : U;                 // a group of four 4x4 TUs waits for what an 8x8 TU in its place would wait for
      // left column and below-left: units (x4 - 1, y4 .. y4 + 2U - 1); the corner (x4 - 1, y4 - 1) belongs to this column too when y4 > 0
      for (int y = (y4 > 0 ? y4 - 1 : 0); y < min(pw, y4 + 2 * Un); y++)
        if (x4 == 0 || z_of(x4 - 1, y) < e.z) nc |= 1u << y;
      // row above and above-right: units (x4 - 1 .. x4 + 2U - 1, y4 - 1); for y4 = 0 the row of the CTUs above, corner included
      if (y4 == 0) { for (int c = x4 - 1; c < x4 + 2 * Un; c++) nr |= 1ull << (c + 1); }
      else for (int c = x4; c < min(pw, x4 + 2 * Un); c++) if (z_of(c, y4 - 1) < e.z) nr |= 1ull << (c + 1);
    }
    // availability of the TU's reference units (intra_tu: 2U units of the left column bottom-up, the corner, 2U units of the row above)
    uint64_t am = 0;
    if (e.kind != 1) {
      const int lx = ctu_x + 4 * x4, ly = ctu_y + 4 * y4;
      for (int u = 0; u <= 4 * U; u++) {
        int px, py;
        if (u < 2 * U) { px = lx - 4; py = ly + 4 * (2 * U - 1 - u); }
        else if (u == 2 * U) { px = lx - 4; py = ly - 4; }
        else { px = lx + 4 * (u - 2 * U - 1); py = ly - 4; }
#if defined(INTRA_EXP) && (INTRA_EXP & 4)      // experiment (wrong samples): every unit counts as available
        if (px >= 0 && py >= 0) am |= 1ull << u;
#else
        if (intra_avail(P, ctu, e.z, px, py, cip != 0, (unsigned)nb_same, L.m_pred)) am |= 1ull << u;
#endif
      }
    }
    // fillReferenceSamples pads unavailable samples from the nearest available one before them (the first available one for those in front):
    // when the available units are one run -- missing below-left and / or above-right units, the usual case -- that is a clamp of the index
    uint32_t cl = 0xffffffffu;
    if (am != 0) {
      const int j0 = __builtin_ctzll(am), j1 = 63 - __builtin_clzll(am);
      if ((am >> j0) == (2ull << (j1 - j0)) - 1ull) {
        const int us = 4 >> cs, nn = U * us, corner = 2 * nn;      // samples per unit, TU size, index of the corner sample
        const int lo = j0 < 2 * U ? j0 * us : (j0 == 2 * U ? corner : corner + 1 + (j0 - 2 * U - 1) * us);
        const int hi = j1 < 2 * U ? j1 * us + us - 1 : (j1 == 2 * U ? corner : corner + 1 + (j1 - 2 * U - 1) * us + us - 1);
        cl = (uint32_t)lo | ((uint32_t)hi << 16);
      }
    }
    Q.put(i, nc, nr, am, cl, tu_desc(e));
  }
  wave_lds_sync();
#if defined(INTRA_STOP) && INTRA_STOP == 5
  if (IntraLds::lean && pw > 0) return;
#endif
    // ---- 3. what is final before anything ran (the neighbours may pass inter areas at once), the pending mask
    if (lane < 8) Q.pend[lane] = n_sched >= 32 * (lane + 1) ? 0xffffffffu : (n_sched > 32 * lane ? (1u << (n_sched - 32 * lane)) - 1u : 0u);
    if (lane == 0) Q.n_tus = n_sched;                          // (what the scans look at: the entries behind are reached through Q.member only)
    if (lane < 3) Q.got[lane] = 0;
    wave_lds_sync();
    const uint32_t word0 = Q.done_c[pw - 1] | (Q.done_r[pw - 1] << 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && word0) __hip_atomic_fetch_or(nb.prog + ctu, word0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef INTRA_TIMING
    if (lane == 0) Q.t_c = wall_clock64();
#endif
  }
  __syncthreads();
#if defined(INTRA_STOP) && INTRA_STOP == 3
  if (pw > 0) return;
#endif
  const int n_tus = Q.n_tus;
  uint32_t spins = 0;
  // A list of up to 64 entries is one scan round, lane c = entry c: what does not change while the CTU runs -- the entry, its column and row
  // words -- is read ONCE and stays in the lane's registers; a round then reads the pending word and the two done masks only.
  // (not in the three-wave kernel of batches: sixteen I pictures were 2.5 % slower with it -- idle waves that scan faster -- where one picture gained 2.7 %)
  const bool cached = n_tus <= 64 && (IntraLds::lean || blockDim.x > 192);
  uint32_t my_e = 0, my_col = 0;
  uint64_t my_row = 0;
  if (cached && lane < n_tus) { my_e = __builtin_bit_cast(uint32_t, Q.tu[lane]); my_col = Q.e_col[lane]; my_row = Q.need_row(lane); }
#ifdef INTRA_TIMING
  unsigned long long tm0 = wall_clock64();
#define TM_ADD(field) { const unsigned long long tm1 = wall_clock64(); if (lane == 0) atomicAdd(&Q.field, tm1 - tm0); tm0 = tm1; }
#else
#define TM_ADD(field)
#endif
  if (!sparse && wv == (int)(blockDim.x >> 6) - 1) fetch_border(org, h_pitch, comp, cs, nb.ext_col, nb.ext_row, Q, L);   // what is final next door already
  for (;;) {
    TK_START();
    // the scheduler words in LDS (pend, done_c, done_r, got) are updated by the other waves with atomics: read them afresh in every round
    asm volatile("" ::: "memory");
    // the first list entry that is pending and ready (the list is in priority order); the pending masks may be stale by the time the
    // entry is claimed: the atomic AND decides
    int i = -1;
    bool any_pending = false;
    // (what the lane read about its entry stays in its registers: the winner's words are fetched from the winning LANE -- v_readlane -- instead of
    // from LDS a second time)
    uint32_t sc_e = 0, sc_col = 0;
    uint64_t sc_row = 0;
    int sc_lane = 0;
    for (int wbase = 0; wbase < n_tus && i < 0; wbase += 64) {
      const int c = wbase + lane;
      const uint32_t pm = Q.pend[c >> 5];
      const bool pending = c < n_tus && ((pm >> (c & 31)) & 1);
      bool ready = false;
      if (pending) {
        if (cached) { sc_e = my_e; sc_col = my_col; sc_row = my_row; }
        else { sc_e = __builtin_bit_cast(uint32_t, Q.tu[c]); sc_col = Q.e_col[c]; sc_row = Q.need_row(c); }
        const TuRun e = __builtin_bit_cast(TuRun, sc_e);
        const int x4 = zscan_x(e.z), y4 = zscan_y(e.z);
        const uint32_t have_c = x4 == 0 ? nb.ext_col : Q.done_c[x4 - 1];
        const uint64_t have_r = y4 == 0 ? nb.ext_row : ((uint64_t)Q.done_r[y4 - 1] << 1) | 1ull;
        ready = ((sc_col & 0xffffu) & ~have_c) == 0 && (sc_row & ~have_r) == 0;
      }
      any_pending |= __builtin_amdgcn_ballot_w64(pending) != 0;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(ready);
      if (m) { sc_lane = (int)__builtin_ctzll(m); i = wbase + sc_lane; }
    }
    if (i < 0) {
      if (!any_pending) break;                               // every TU is taken: the waves that hold one finish it
      if (nb.broken) break;
      // nothing this wave can run: either another wave is inside a TU whose units will make one ready, or the neighbours have to get further
      __builtin_amdgcn_s_sleep(4);
      ++spins;
#ifdef INTRA_TIMING
      if (lane == 0) atomicAdd(&Q.n_idle, 1u);
#endif
      // (every fourth round; a count of the waves inside a TU used to decide between every round and every sixteenth -- two LDS atomics per
      // TU on the chain for a distinction the picture does not feel: 2.6 % of an I picture)
      if ((spins & 3) == 0) {
        if (spins > (1u << 22)) { if (lane == 0) atomicOr(P.fault, 1u); nb.broken = true; }
        poll_neighbours(P, nb, pw);
        if (!sparse) fetch_border(org, h_pitch, comp, cs, nb.ext_col, nb.ext_row, Q, L);   // whatever has become final next door
      }
      TM_ADD(t_idle)
      continue;
    }
    TK(5)
    // claim it
    uint32_t old = 0;
    if (lane == 0) {
      old = atomicAnd(&Q.pend[i >> 5], ~(1u << (i & 31)));
    }
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    if (!((old >> (i & 31)) & 1)) continue;                  // another wave was faster
    TK(6)
    // everything about the TU is the same in all lanes, but it comes out of LDS into vector registers: moved to scalar ones, what is derived
    // from it (coordinates, the mode's angle, branch conditions) runs on the scalar unit beside the lanes' own work
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const TuRun e = __builtin_bit_cast(TuRun, (uint32_t)__builtin_amdgcn_readlane((int)sc_e, sc_lane));
    const uint32_t e_colw = (uint32_t)__builtin_amdgcn_readlane((int)sc_col, sc_lane);
    const uint64_t e_roww = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sc_row, sc_lane) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sc_row >> 32), sc_lane) << 32);
    int x4, y4, U;
    footprint(e, x4, y4, U);
    const int z = e.z_cu, zc = e.z;
    if (e.kind == 1) {
      // PCM CU (TDecCu::xReconPCM, TDecCu.cpp:770-830): the transmitted samples, shifted up to the coding bit depth; lane n = row n
      const int n_cu = (1 << e.log2n) >> cs;
      const int sx = (ctu_x + 4 * x4) >> cs, sy = (ctu_y + 4 * y4) >> cs;
      if (lane < n_cu) {
        const int16_t* src = P.pcm[comp] + (size_t)ctu * ((size_t)(1 << (2 * P.log2ctu)) >> (2 * cs)) + ((16 * z) >> (2 * cs)) + lane * n_cu;
        int16_t* prow = h_plane + (ptrdiff_t)(sy + lane) * h_pitch + (comp ? kCStep : 1) * sx;
        uint32_t* row = reinterpret_cast<uint32_t*>(prow);
        uint32_t* lrow = reinterpret_cast<uint32_t*>(&L.pix[sy - (ctu_y >> cs) + lane][sx - (ctu_x >> cs) + 2]);
        for (int x = 0; x < n_cu; x += 2) {
          const uint32_t v = ldg(reinterpret_cast<const uint32_t*>(src + x));
          const uint32_t o = ((v & 0xffffu) << P.pcm_shift[comp]) | ((v >> 16) << (16 + P.pcm_shift[comp]));
          if (!sparse) lrow[x / 2] = o;                       // (no copy of the CTU in LDS otherwise -- the LEAN kernel has none at all)
          if (comp == 0) st_coh(row + x / 2, o); else st_coh_c2(prow + kCStep * x, o);
        }
      }
    } else {
      // one TU of the list: entry idx (its availability), first partition zs, origin unit (xs, ys) of the CTU
      auto run_tu = [&](int idx, int zs, int xs, int ys, uint32_t colw) {
        TuCtx t;
        t.comp = comp; t.ctu = ctu; t.z_tu = zs; t.cip = cip; t.slice = slice; t.tile = tile; t.nb_same = nb_same;
        t.cx0 = ctu_x >> cs; t.cy0 = ctu_y >> cs;
        t.x0 = (ctu_x + 4 * xs) >> cs; t.y0 = (ctu_y + 4 * ys) >> cs;
        t.log2n = e.log2n;
        t.plane = h_plane; t.pitch = h_pitch; t.bd = h_bd; t.log2ctu = h_log2ctu; t.rext = h_rext; t.strong = h_strong; t.cs = cs; t.fmt = h_fmt;
        t.sparse = sparse; t.resid = h_resid; t.rtw = h_rtw;
        t.mode = (int)((colw >> 24) & 63u);                          // (the TU's descriptor in bits 24-31 of its column word: IntraSched::put)
        {
          const uint64_t a = Q.avail(idx);
          t.am = (unsigned long long)(uint32_t)uni((int)(uint32_t)a) | ((unsigned long long)(uint32_t)uni((int)(uint32_t)(a >> 32)) << 32);
          const uint32_t cl = (uint32_t)uni((int)Q.clampi(idx));
          t.sub_lo = cl == 0xffffffffu ? -1 : (int)(cl & 0xffff); t.sub_hi = (int)(cl >> 16);
        }
        t.cbf = (int)((colw >> 30) & 1u);
        t.bypass = (int)(colw >> 31);
        TK(7)
        intra_tu_any(P, t, L, W);
#ifdef INTRA_TIMING2
        tk0 = clock64();
#endif
      };
      if (!sparse && (x4 == 0 || y4 == 0))                      // (a TU inside the CTU reads nothing from next door; a group: what an 8x8 TU would read)
        fetch_border(org, h_pitch, comp, cs, x4 == 0 ? (e_colw & 0xffffu) : 0u, y4 == 0 ? e_roww : 0ull, Q, L);
#ifdef INTRA_TIMING
      if (lane == 0) { atomicMin(&Q.t_first, wall_clock64()); atomicAdd(&Q.n_run, 1u); }
#endif
      TM_ADD(t_claim)
      {
        // one TU -- or the four 4x4 luma TUs of the area, in z order (each predicts from the ones before it; intra_tu ends with the hand-off that
        // puts its samples into the CTU copy).  The area's units are published together: whatever waits for one of them -- the TUs to the right,
        // below and below-left -- waits for the fourth TU too.  (One call site: the TU code is the bulk of this kernel's 47 KB.)
        const int last = e.kind == 2 ? 3 : 0;
        const uint32_t mem = e.kind == 2 ? (uint32_t)uni((int)*reinterpret_cast<const uint32_t*>(Q.member[zc >> 2])) : (uint32_t)i;
        for (int j = 0; j <= last; j++) {
          const int idx = (int)((mem >> (8 * j)) & 0xff);
          // (a group's other three TUs: their own entries' words)
          run_tu(idx, zc + j, x4 + (j & 1), y4 + (j >> 1), j == 0 ? e_colw : (uint32_t)uni((int)Q.e_col[idx]));
        }
        if (last) U = 2;                                     // the common exit publishes the 2 x 2 units of the area
      }
      TM_ADD(t_tu)
    }
    if (sparse) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (sparse CTU: the samples are in the picture, where the CTU's other TUs read them)
    wave_lds_sync();                                         // the TU's samples are in the CTU copy before its units count as final
    TK(8)
    mark_done(x4, y4, U);
    if (x4 + U == pw || y4 + U == pw) {
      // the neighbours are shown THIS TU's border units, once its stores are acknowledged
      const uint32_t bits = (x4 + U == pw ? ((1u << U) - 1u) << y4 : 0u) | (y4 + U == pw ? (((1u << U) - 1u) << x4) << 16 : 0u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_or(nb.prog + ctu, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef INTRA_TIMING
    if (lane == 0) atomicMax(&Q.t_last, wall_clock64());
#endif
    TM_ADD(t_post)
    TK(9)
#ifdef INTRA_TIMING2
    if (lane == 0) W.tm[15] += 1u;
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// grid: x = (picture of the batch, component), y = position in `order`; one wave per block.  A CTU that holds intra CUs waits
// for its left, above-left, above and above-right neighbours -- those of them that hold intra CUs themselves and belong to
// this call -- and nothing else: intra CUs scattered over a P picture are reconstructed side by side, an I picture
// degenerates into the classic CTU wavefront.  `order` lists the CTUs by anti-diagonal (2*row + column): all four
// neighbours lie on earlier diagonals, so every block a block can wait for has a smaller linear index and was dispatched
// before it (no deadlock however few blocks are resident), and the blocks resident at any time are the wavefront itself.
// LEAN: every CTU on the unstaged path (below), in a third of the LDS: the variant for calls without I slices
template <int WAVES, bool LEAN>          // waves per CTU and component: they run the CTU's ready TUs side by side (intra_ctu)
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(LEAN ? INTRA_LEAN_OCC : (WAVES <= 4 ? INTRA_MANY_OCC : 1)))) k_intra(const PicDev* __restrict__ pics, Batch b, const int32_t* __restrict__ order) {
  __shared__ IntraLdsT<LEAN> L;
#ifdef INTRA_LDS_PAD                     // experiment: fewer workgroups per CU
  __shared__ uint32_t lds_pad[INTRA_LDS_PAD / 4];
  if (blockIdx.x == 0xffffffffu) lds_pad[threadIdx.x] = 1;
#endif
  __shared__ IntraSchedT<LEAN> Q;
  __shared__ IntraScratch W[WAVES];
  const int slot = blockIdx.x / 3, comp = blockIdx.x % 3, ctu = ldg(order + blockIdx.y);
  const PicDev& P = pics[b.pic[slot]];
  const int first = b.first_ctu[slot], last = first + b.num_ctus[slot] - 1;
  if (!P.has_intra_dir || ctu < first || ctu > last || !ldg(P.ctu_intra + ctu) || (comp && (P.mono || P.fmt == 2))) return;      // (4:2:2 chroma: k_intra_chroma_422)
  if (ctu == P.debug_skip_ctu) return;                         // (test hook: its neighbours run into the bounded wait)
#if defined(INTRA_EXP) && (INTRA_EXP & 16)    // experiment (no chroma): the luma wavefront alone
  if (comp) return;
#endif
  uint32_t* done = P.intra_done + (size_t)comp * P.num_ctus;
  const int cx = ctu % P.ctus_w, cy = ctu / P.ctus_w;
#ifdef INTRA_TIMING
  const unsigned long long tk0 = wall_clock64();
  if (threadIdx.x == 0) { Q.t_tu = Q.t_claim = Q.t_post = Q.t_idle = 0; Q.t_first = ~0ull; Q.t_last = 0; Q.n_run = Q.n_idle = 0; }
#endif
  const int nb[4] = {cx > 0 ? ctu - 1 : -1, (cx > 0 && cy > 0) ? ctu - P.ctus_w - 1 : -1, cy > 0 ? ctu - P.ctus_w : -1,
                     (cy > 0 && cx + 1 < P.ctus_w) ? ctu - P.ctus_w + 1 : -1};
  // which borders matter: an intra CU reads neighbouring samples only across a border it touches, and only the samples of the
  // neighbour's intra CUs are still being written.  One dword of pred_mode per lane = four z-consecutive partitions = one 8x8 area.
  const int lane = threadIdx.x & 63, pw = P.pw;
  auto border_mask = [&](int c, unsigned& left, unsigned& right, unsigned& top, unsigned& bottom) {
    // bit masks over 8x8 areas of CTU c that hold an intra partition in the first / last column or row of the CTU
    bool l = false, r = false, t = false, bm = false;
    if (4 * lane < P.parts) {
      const uint32_t pm = ldg(reinterpret_cast<const uint32_t*>(P.pred_mode + (size_t)c * P.parts) + lane);
      const int x = zscan_x(4 * lane), y = zscan_y(4 * lane);               // partition coordinates of the area's first partition
      const bool i0 = (pm & 0xff) == HMGPU_MODE_INTRA, i1 = ((pm >> 8) & 0xff) == HMGPU_MODE_INTRA,
                 i2 = ((pm >> 16) & 0xff) == HMGPU_MODE_INTRA, i3 = (pm >> 24) == HMGPU_MODE_INTRA;      // (x,y) (x+1,y) (x,y+1) (x+1,y+1)
      l = x == 0 && (i0 || i2); r = x + 2 == pw && (i1 || i3); t = y == 0 && (i0 || i1); bm = y + 2 == pw && (i2 || i3);
    }
    left = __builtin_amdgcn_ballot_w64(l) != 0; right = __builtin_amdgcn_ballot_w64(r) != 0;
    top = __builtin_amdgcn_ballot_w64(t) != 0; bottom = __builtin_amdgcn_ballot_w64(bm) != 0;
  };
  // the residual of this CTU does not depend on anybody: on its way into LDS while the neighbours finish (all threads of the workgroup
  // stage; the barrier in front of the TU loop orders it).  Tiles of TUs that are not coded hold stale data and are never read.
  // A CTU with few intra CUs (scattered ones in a P picture) does not stage at all: 16 KB of samples and residual per CTU and component
  // for a handful of TUs was most of what such a picture's intra CUs cost.  Its TUs take their reference samples and residual from the
  // picture (intra_tu), one round trip each on a chain of a few TUs.
  const bool sparse = LEAN || (int)ldg(P.ctu_intra + ctu) * 64 <= INTRA_SPARSE_MAX * (P.parts >> 2);
  const int wv = threadIdx.x >> 6;
  {
  if (!sparse) {
    const int cs = (comp && P.csx) ? 1 : 0;
    const int tpr = ((1 << P.log2ctu) >> cs) >> 3, rtw = (P.grid_w / 2) >> cs;           // tiles per CTU row / per picture row
    const int tx0 = cx * tpr, ty0 = cy * tpr;
    const int16_t* src = P.resid[comp];
    for (int i = threadIdx.x; i < tpr * tpr * 8; i += blockDim.x) {                       // 16-byte vectors, eight per tile
      const int tile = i >> 3, v = i & 7;
      *reinterpret_cast<u32x4*>(&L.res[tile * 64 + v * 8]) = ldg4(src + ((size_t)(ty0 + tile / tpr) * rtw + tx0 + tile % tpr) * 64 + v * 8);
    }
  }
  intra_stage(P, comp, ctu, L, !sparse);
  }
#ifdef INTRA_TIMING
  if (threadIdx.x == 0) Q.t_a = wall_clock64();
#endif
#if defined(INTRA_STOP) && INTRA_STOP == 1      // experiment (no reconstruction): the time of the launch up to here
  if (pw > 0) return;
#endif
  unsigned my_l, my_r, my_t, my_b;
  border_mask(ctu, my_l, my_r, my_t, my_b);
  Neighbours nbs;
  nbs.prog = done; nbs.broken = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int n = nb[k];
    nbs.ctu[k] = n; nbs.wait[k] = false;
    if (n < first || !ldg(P.ctu_intra + n)) continue;      // outside the picture / finished by an earlier call / no intra CUs: complete already
    unsigned n_l, n_r, n_t, n_b;
    border_mask(n, n_l, n_r, n_t, n_b);
    // left: my first column against its last column; above-left: the corner (covered by column and row tests, conservatively);
    // above and above-right: my first row against its last row
    nbs.wait[k] = k == 0 ? (my_l && n_r) : k == 1 ? ((my_l || my_t) && n_r && n_b) : (my_t && n_b);
#ifdef INTRA_NOWAIT
    nbs.wait[k] = false;                                     // experiment: no CTU waits for another (wrong samples, the time of the work alone)
#endif
  }
  // The waits (intra_ctu's polling) cannot deadlock as long as workgroups are dispatched in linear order (the neighbours lie on earlier
  // anti-diagonals, i.e. at smaller block indices: they are resident or finished when this block runs, and never wait for this one).
  // HIP does not promise that order, so every spin is bounded: a block that gives up flags the picture (the host reports HMGPU_EDEVICE
  // at its next sync) instead of hanging the device.
  poll_neighbours(P, nbs, P.pw);
  wave_lds_sync();
#if defined(INTRA_STOP) && INTRA_STOP == 2
  if (pw > 0) return;
#endif
#ifdef INTRA_TIMING
  if (threadIdx.x == 0) Q.t_b = wall_clock64();
#endif
#ifdef INTRA_TIMING2
  if ((threadIdx.x & 63) < 16) W[wv].tm[threadIdx.x & 63] = 0;
  wave_lds_sync();
#endif
  intra_ctu(P, comp, ctu, sparse, L, Q, W[wv], nbs);
  __syncthreads();                                           // every wave's stores are acknowledged (intra_ctu ends with the wait)
#ifdef INTRA_TIMING
  if (threadIdx.x == 0 && comp == 0 && slot == 0 && (cy == 10 || cy == 11) && cx >= 20 && cx < 32)
    printf("TM cy %d cx %d start %llu first %llu last %llu end %llu ntu %u tu %llu claim %llu post %llu idle %llu nidle %u stage %llu nbrs %llu list %llu\n", cy, cx, tk0, Q.t_first, Q.t_last,
           wall_clock64(), Q.n_run, Q.t_tu, Q.t_claim, Q.t_post, Q.t_idle, Q.n_idle, Q.t_a - tk0, Q.t_b - Q.t_a, Q.t_c - Q.t_b);
#endif
#ifdef INTRA_TIMING2
  if (threadIdx.x == 0 && comp == 0 && slot == 0 && (cy == 10 || cy == 11) && cx >= 20 && cx < 32) {
    unsigned sum[16];
    for (int k = 0; k < 16; k++) { sum[k] = 0; for (int w = 0; w < WAVES; w++) sum[k] += W[w].tm[k]; }
    printf("TK cy %d cx %d tus %u line %u filt %u pred %u resid %u store %u scan %u claim %u setup %u sync %u post %u\n", cy, cx, sum[15], sum[0], sum[1], sum[2], sum[3], sum[4],
           sum[5], sum[6], sum[7], sum[8], sum[9]);
  }
#endif
  // (No release fence in front of the last word: every store to the picture is a write-through atomic store that has been acknowledged by now --
  // intra_ctu ends with the wait.  An agent-scope release fence writes the L2 back: one per workgroup, ~98 000 of them in sixteen 2160p P
  // pictures with scattered intra CUs, was HALF of this kernel's time there -- 0.65 of 1.33 ms, found by cutting the kernel short phase by phase.)
  if (wv == 0) publish_progress(done, ctu, 0xffffffffu);
}

// ---- 4:2:2 chroma.  A chroma block of a transform unit is two SQUARES of half the unit's width, one above the other, the
// lower one predicted from the upper one's reconstruction (xIntraRecBlk's TComTU::VERTICAL_SPLIT, TDecCu.cpp:505-520); a 4x4 luma unit
// covers 2 x 4 chroma samples, so the TU scheduler of intra_ctu (square footprints on the unit grid) does not describe it.  Correct first:
// ONE wave per picture and component walks the CTUs in raster order and their transform units in z order -- decoding order, so every
// reference sample is final when it is read (luma and the other component are other kernels' / waves' business: chroma never reads them).
// Samples go through the coherent level like k_intra's (the wave's own earlier stores must be what its later loads see).  No smoothing and
// no edge filters: both are luma-only at this format (filterIntraReferenceSamples, TComChromaFormat.h:150-153; TComPrediction.cpp:275).
namespace {
__constant__ uint8_t c_mode422[36] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31, 36};

__device__ inline bool avail_glob(const PicDev& P, int ctu, int z_tu, int px, int py, bool cip, int slice, int tile) {
  if (px < 0 || py < 0 || px >= P.width || py >= P.height) return false;
  const int m = (1 << P.log2ctu) - 1;
  const int nctu = (py >> P.log2ctu) * P.ctus_w + (px >> P.log2ctu);
  const int nz = z_of((px & m) >> 2, (py & m) >> 2);
  if (nctu == ctu) { if (nz >= z_tu) return false; }
  else {
    if (nctu > ctu) return false;
    if (ldg(P.slice_idx + nctu) != slice || ldg(P.tile_idx + nctu) != tile) return false;
  }
  return !cip || ldg(P.pred_mode + (size_t)nctu * P.parts + nz) == HMGPU_MODE_INTRA;
}

// one square of n x n chroma samples of component comp whose first 4x4 luma partition is zb; z_mode: the partition the mode is stored at
__device__ void intra_square_422(const PicDev& P, int comp, int ctu, int zb, int z_mode, int log2n, int slice, int tile, bool cip, int* line, int* proj) {
  const int lane = threadIdx.x & 63, n = 1 << log2n, corner = 2 * n, total = 4 * n + 1;
  const int ctu_x = (ctu % P.ctus_w) << P.log2ctu, ctu_y = (ctu / P.ctus_w) << P.log2ctu;
  const int lx = ctu_x + 4 * zscan_x(zb), ly = ctu_y + 4 * zscan_y(zb);           // luma position of the square
  const int x0 = lx >> 1, y0 = ly;
  const int pitch = P.pitch[1], bd = P.bd[comp], maxv = (1 << bd) - 1;
  int16_t* plane = P.rec[comp];
  int mode = ldg(P.intra_dir[1] + (size_t)ctu * P.parts + z_mode);
  if (mode == 36) {
    const int cu_parts = P.parts >> (2 * ldg(P.depth + (size_t)ctu * P.parts + z_mode));
    mode = ldg(P.intra_dir[0] + (size_t)ctu * P.parts + (z_mode & ~(cu_parts - 1)));
  }
  mode = c_mode422[mode];
  // ---- reference line: [0, 2n) the left column bottom-up, [2n] the corner, (2n, 4n] the row above; a 4x4 luma partition is 4 rows of the
  // column and 2 samples of the row.  -1 = not available
  for (int i = lane; i < total; i += 64) {
    int px, py, sx, sy;
    if (i < corner) { const int r = corner - 1 - i; px = lx - 4; py = ly + r; sx = x0 - 1; sy = y0 + r; }
    else if (i == corner) { px = lx - 4; py = ly - 4; sx = x0 - 1; sy = y0 - 1; }
    else { const int c = i - corner - 1; px = lx + 2 * c; py = ly - 4; sx = x0 + c; sy = y0 - 1; }
    int v = -1;
    if (avail_glob(P, ctu, zb, px, py, cip, slice, tile)) {
      const uint32_t w = ld_coh(reinterpret_cast<const uint32_t*>(P.rec[1] + (ptrdiff_t)sy * pitch + kCStep * sx));      // the (Cb, Cr) pair
      v = (int)(comp == 1 ? (w & 0xffffu) : (w >> 16));
    }
    line[i] = v;
  }
  wave_lds_sync();
  if (lane == 0) {
    // fillReferenceSamples (TComPattern.cpp:336-478): unavailable samples take the nearest available one before them, the ones in front the first available
    int first = 0;
    while (first < total && line[first] < 0) first++;
    if (first == total) { for (int i = 0; i < total; i++) line[i] = 1 << (bd - 1); }
    else {
      for (int i = 0; i < first; i++) line[i] = line[first];
      for (int i = first + 1; i < total; i++) if (line[i] < 0) line[i] = line[i - 1];
    }
  }
  wave_lds_sync();
  // ---- prediction of row `lane` (predIntraAng / xPredIntraPlanar, TComPrediction.cpp:245-491, 746-800)
  const int row = lane;
  const bool active = lane < n;
  int p[16];
  if (mode == 0) {
    const int left = line[corner - 1 - min(row, n - 1)], bl = line[corner - 1 - n], tr = line[corner + 1 + n];
    for (int x = 0; x < n; x++) { const int ab = line[corner + 1 + x]; p[x] = ((left << log2n) + n + (x + 1) * (tr - left) + (ab << log2n) + (row + 1) * (bl - ab)) >> (log2n + 1); }
  } else if (mode == 1) {
    int sum = n;
    for (int x = 0; x < n; x++) sum += line[corner + 1 + x] + line[corner - 1 - x];
    for (int x = 0; x < n; x++) p[x] = sum >> (log2n + 1);
  } else {
    const bool ver = mode >= 18;
    const int am_ = ver ? mode - 26 : -(mode - 10), aa = abs(am_);
    // intraPredAngle {0, 2, 5, 9, 13, 17, 21, 26, 32} and its inverse {-, 4096, 1638, 910, 630, 482, 390, 315, 256} (TComPrediction.cpp:276-277) out of
    // packed constants: the chains of conditionals the tables were written as before compiled into ~20 scalar branches on the TU chain
    const int ang_abs = aa >= 8 ? 32 : (int)((0x1a15110d09050200ull >> (8 * aa)) & 0xffu);
    const int inv = aa >= 8 ? 256 : (int)(((aa & 4) ? 0x013b018601e20276ull : 0x038e066610000000ull) >> (16 * (aa & 3)) & 0xffffu);
    const int ang = am_ < 0 ? -ang_abs : ang_abs, sgn = ver ? 1 : -1;        // MAIN(i) = line[corner + sgn * i], SIDE(i) = line[corner - sgn * i]
    // main reference with its extension to negative indices: proj[k + 32], k in [-n, 2n]
    for (int k = lane - n; k <= 2 * n; k += 64) {
      int v = 0;
      if (k >= 0) v = line[corner + sgn * k];
      else if (ang < 0 && k > ((n * ang) >> 5)) v = line[corner - sgn * ((128 + (-k) * inv) >> 8)];
      proj[k + 32] = v;
    }
    wave_lds_sync();
    const int* r = proj + 32;
    for (int x = 0; x < n; x++) {
      // vertical modes: row y = lane, column x; horizontal ones are the transpose
      const int yy = ver ? min(row, n - 1) : x, xx = ver ? x : min(row, n - 1);
      const int pos = (yy + 1) * ang, di = pos >> 5, df = pos & 31;
      p[x] = df ? ((32 - df) * r[xx + di + 1] + df * r[xx + di + 2] + 16) >> 5 : r[xx + di + 1];
    }
  }
  // ---- residual (k_itx's tiles; zero where nothing is coded) + reconstruction
  if (active) {
    const int rtw = (P.grid_w / 2) >> 1, y = y0 + row;
    const int16_t* rrow = P.resid[comp] + ((size_t)((y >> 3) * rtw + (x0 >> 3)) * 8 + resid_slot(y)) * 8 + (x0 & 7);
    int16_t* prow = plane + (ptrdiff_t)y * pitch + kCStep * x0;
    for (int x = 0; x < n; x += 2) {
      const uint32_t rs = ldg(reinterpret_cast<const uint32_t*>(rrow + (x & 7) + (x >> 3) * 64));
      const int v0 = clip3(0, maxv, p[x] + (int)(int16_t)(rs & 0xffffu)), v1 = clip3(0, maxv, p[x + 1] + ((int)rs >> 16));
      st_coh_c2(prow + kCStep * x, (uint32_t)v0 | ((uint32_t)v1 << 16));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_lds_sync();
}
}  // namespace

__global__ void __launch_bounds__(64) k_intra_chroma_422(const PicDev* __restrict__ pics, Batch b) {
  __shared__ int line[4 * 16 + 4], proj[3 * 16 + 36];
  const int slot = blockIdx.x >> 1, comp = 1 + (blockIdx.x & 1);
  const PicDev& P = pics[b.pic[slot]];
  if (!P.has_intra_dir) return;
  const int lane = threadIdx.x & 63;
  for (int ctu = b.first_ctu[slot]; ctu < b.first_ctu[slot] + b.num_ctus[slot]; ctu++) {
    if (!ldg(P.ctu_intra + ctu)) continue;
    const int slice = ldg(P.slice_idx + ctu), tile = ldg(P.tile_idx + ctu);
    const bool cip = ldg(&P.slices[slice].constrained_intra_pred) != 0;
    const int ctu_x = (ctu % P.ctus_w) << P.log2ctu, ctu_y = (ctu / P.ctus_w) << P.log2ctu;
    for (int z = 0; z < P.parts;) {
      const size_t i = (size_t)ctu * P.parts + z;
      const int px = ctu_x + 4 * zscan_x(z), py = ctu_y + 4 * zscan_y(z);
      if (px >= P.width || py >= P.height || (int)ldg(P.part_size + i) == HMGPU_SIZE_NONE || ldg(P.pred_mode + i) != HMGPU_MODE_INTRA) { z++; continue; }
      const int depth = ldg(P.depth + i), log2cu = P.log2ctu - depth, cu_parts = 1 << (2 * (log2cu - 2));
      if (ldg(P.ipcm + i) && P.pcm[comp] != nullptr) {
        // PCM CU (xReconPCM, TDecCu.cpp:770-830): (cu / 2) x cu transmitted samples, shifted up to the coding bit depth; lane = row
        const int cw = (1 << log2cu) >> 1, chh = 1 << log2cu;
        const int16_t* src = P.pcm[comp] + (size_t)ctu * ((size_t)(1 << (2 * P.log2ctu)) >> 1) + 8 * z;
        for (int r = lane; r < chh; r += 64)
          for (int x = 0; x < cw; x += 2) {
            const uint32_t v = ldg(reinterpret_cast<const uint32_t*>(src + r * cw + x));
            st_coh_c2(P.rec[comp] + (ptrdiff_t)(py + r) * P.pitch[1] + kCStep * ((px >> 1) + x),
                      ((v & 0xffffu) << P.pcm_shift[comp]) | ((v >> 16) << (16 + P.pcm_shift[comp])));
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        z += cu_parts;
        continue;
      }
      // the transform unit node that carries a chroma block here: the luma TU, or the 8x8 area of four 4x4 luma TUs
      const int log2tu = max(log2cu - (int)ldg(P.tr_idx + i), 3), nparts = 1 << (2 * (log2tu - 2));
      intra_square_422(P, comp, ctu, z, z, log2tu - 1, slice, tile, cip, line, proj);
      intra_square_422(P, comp, ctu, z + nparts / 2, z, log2tu - 1, slice, tile, cip, line, proj);
      z += nparts;
    }
  }
}

void launch_intra_chroma_422(const PicDev* pics, const Batch& b, hipStream_t s) {
  hipLaunchKernelGGL(k_intra_chroma_422, dim3((unsigned)b.n * 2), dim3(64), 0, s, pics, b);
}

void launch_intra(const PicDev* pics, const Batch& b, const int32_t* order, int num_ctus, bool lean, hipStream_t s) {
  // measured on 2160p I pictures: one picture 17.9 / 11.6 / 9.7 ms with 1 / 2 / 4 waves per CTU, sixteen at once 19.4 / 13.5 / 17.7;
  // with the residual from k_itx (140 instead of 212 VGPRs: three waves per SIMD) 8.7 ms (4 or 6 waves), sixteen 11.4 / 10.8 / 11.7 (2 / 3 / 4);
  // after the trims of the TU chain: 7.78 / 7.41 / 7.27 / 7.15 ms with 3 / 4 / 6 / 8 waves, sixteen 9.93 / 9.66 / 10.7 (2 / 3 / 4)
#ifndef INTRA_WAVES_ONE
#define INTRA_WAVES_ONE 8
#define INTRA_WAVES_MANY 3
#endif
  // Many pictures without I slices (P / B pictures with scattered intra CUs): independent CTUs in plenty, each with a handful of TUs.  What counts
  // is how many of them a CU holds at a time: one wave per CTU and no staging (LEAN), a third of the LDS and of the wave slots of the general kernel
  // -- 2160p x 16 with 5 / 10 / 25 % intra CUs: see DESIGN.md 4.12 for the two kernels side by side.
#ifndef INTRA_WAVES_LEAN
#define INTRA_WAVES_LEAN 1
#endif
  const dim3 grid((unsigned)b.n * 3, (unsigned)num_ctus);
#ifdef INTRA_NO_LEAN                     // experiment: the general kernel for every call
  lean = false;
#endif
  if (b.n >= 4 && lean) hipLaunchKernelGGL((k_intra<INTRA_WAVES_LEAN, true>), grid, dim3(64 * INTRA_WAVES_LEAN), 0, s, pics, b, order);
  else if (b.n >= 4) hipLaunchKernelGGL((k_intra<INTRA_WAVES_MANY, false>), grid, dim3(64 * INTRA_WAVES_MANY), 0, s, pics, b, order);
  else hipLaunchKernelGGL((k_intra<INTRA_WAVES_ONE, false>), grid, dim3(64 * INTRA_WAVES_ONE), 0, s, pics, b, order);
}

}  // namespace hmgpu
