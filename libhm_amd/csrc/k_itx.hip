// k_itx.hip -- de-quantisation + inverse transform + reconstruction add for coded TUs.
//   TComTrQuant::invTransformNxN -> xDeQuant (flat) -> xIT -> xITrMxN / xITransformSkip   TComTrQuant.cpp:1423,1203,1836,894,1920
//   TComYuv::addClip (recon = ClipBD(pred + resid))                                      TComYuv.cpp:264
//
// One N-lane group per N x N TU (N = 4, 8, 16, 32; a wave holds 64/N TUs).  Stage 1: lane n owns coefficient column
// n (coalesced 2-byte reads along the row), de-quantises it and runs the N-point 1-D inverse transform in registers
// with the even/odd decomposition (full-rate 24-bit integer multiply-adds, all matrix entries are immediates).  The
// 16-bit-clipped intermediates cross to the row owners through a padded LDS tile (conflict-free b32 writes / b128
// reads).  Stage 2: lane y owns row y, transforms it, adds it to the prediction already sitting in the picture
// (written by the MC kernels) and writes the clipped reconstruction back: the residual never touches HBM.
// TUs come from the per-class lists built by k_prep; blocks stride over a list whose length only the device knows.
#include "hmgpu_dev.h"
#include "itx_core.h"
#include <algorithm>

namespace hmgpu {

// =====================================================================================================================
// Packed 16-bit formulation of the TU pipeline (inter residuals; the intra kernel and the DST keep itx_core.h's 32-bit one).
//
// Everything that crosses a stage boundary in HM's inverse transform is a 16-bit quantity: levels, de-quantised coefficients
// (clipped to 16 bits, TComTrQuant.cpp:1284-1311), first-stage outputs (clipped, :898-901) and the residual.  So
//   * de-quantisation is ONE v_pk_mad_i16 with clamp per two levels whenever the flat scaling is a left shift (every QP the
//     encoder configurations can produce at bit depth <= 10 for N >= 8; a 32-bit path covers the right-shift cases): the
//     saturating multiply IS clip16(level * (scale << shift));
//   * the even/odd butterflies are v_dot2_i32_i16 over pairs of inputs -- half the multiplies of the scalar form.  The pairs
//     the decomposition wants are (1,3)(5,7).. for the odd part, (2,6)(10,14).. for the odd part of the even part, and so on
//     down to (N/2,0): the lane that owns a row writes its 16-bit elements to LDS in exactly that "slot" order, so the lane
//     that owns a column reads ready-made pairs with ds_read_b128 and no register shuffling;
//   * rounding shifts end in v_cvt_pk_i16_i32 (saturating pack = the 16-bit clip), the reconstruction add is a saturating
//     packed add followed by a packed clip to the sample range.
// One 16-bit LDS tile of N x (N+8) per TU serves both hand-offs.
// =====================================================================================================================
namespace {
typedef short short2v __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr uint32_t pkc(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// input index k that lives at slot s of an n_size-point transform: all odd k, then the odd multiples of 2, of 4, ... , n_size/2, 0
__host__ __device__ constexpr int slot_k(int n_size, int s) {
  int size = n_size / 2, g = 1, base = 0;
  while (size >= 1) {
    if (s < base + size) return (2 * (s - base) + 1) << (g - 1);
    base += size; size /= 2; g++;
  }
  return 0;
}
__host__ __device__ constexpr int slot_of_c(int n_size, int k) {
  for (int s = 0; s < n_size; s++) if (slot_k(n_size, s) == k) return s;
  return -1;
}
__device__ inline int slot_of(int n_size, int k) {          // run-time form of the same map
  if (k == 0) return n_size - 1;
  const int g = __builtin_ctz(k) + 1;
  return n_size - (n_size >> (g - 1)) + (k >> g);
}

__device__ inline int dot2c(uint32_t pair, uint32_t taps, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pair), __builtin_bit_cast(short2v, taps), acc, false);
}
// first product of a chain (three-address form: no v_mov to seed the accumulator)
__device__ inline int dot2_z(uint32_t pair, uint32_t taps) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(pair), "s"(taps));
  return d;
}
__device__ inline int dot2_v(uint32_t pair, uint32_t taps, int seed) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pair), "s"(taps), "v"(seed));
  return d;
}
__device__ inline uint32_t pk_mad_sat(uint32_t a, uint32_t b) {                       // per half: sat16(a * b)
  uint32_t d;
  asm("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ inline uint32_t pk_add_sat(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ inline uint32_t cvt_pk_sat(int lo, int hi) {                               // (sat16(lo), sat16(hi))
  uint32_t d;
  asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
  return d;
}
__device__ inline uint32_t pk_clip_u(uint32_t v, uint32_t hi) {                       // per half: min(max(v, 0), hi)
  const short2v z = {0, 0};
  const short2v r = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(short2v, v), z), __builtin_bit_cast(short2v, hi));
  return __builtin_bit_cast(uint32_t, r);
}

// N-point inverse DCT of one column whose inputs arrive as N/2 pairs in slot order (p[OFF..]); out[m] = seed + sum_k T_N[k][m] in[k]
// (partialButterflyInverseN's even/odd structure, TComTrQuant.cpp:468-828)
template <int N, int OFF, int NP>
__device__ inline void idct_pk(const uint32_t (&p)[NP], int seed, int (&out)[N]) {
  if constexpr (N == 2) {
    out[0] = dot2_v(p[OFF], pkc(64, 64), seed);             // pair = (in[1], in[0])
    out[1] = dot2_v(p[OFF], pkc(-64, 64), seed);
  } else {
    int e[N / 2];
    idct_pk<N / 2, OFF + N / 4, NP>(p, seed, e);
#pragma unroll
    for (int m = 0; m < N / 2; m++) {
      int o = dot2_z(p[OFF], pkc(tmat(N, 1, m), tmat(N, 3, m)));
#pragma unroll
      for (int i = 1; i < N / 4; i++) o = dot2c(p[OFF + i], pkc(tmat(N, 4 * i + 1, m), tmat(N, 4 * i + 3, m)), o);
      out[m] = e[m] + o;
      out[N - 1 - m] = e[m] - o;
    }
  }
}

template <int LOG2N> struct PkCfg {
  static constexpr int N = 1 << LOG2N;
  static constexpr int S = N == 4 ? 4 : (N == 8 ? 24 : N + 8);      // int16 per LDS row: odd multiple of the read width (conflict-free ds_read_b128 / b64)
  static constexpr int TU_ELEMS = N * S + (N >= 16 ? 8 : 0);        // + 16 bytes so that the TUs of a wave start in different banks
  static constexpr int TPB = 256 / N;
};

// 16-bit element x of a lane's packed row -> LDS position [x][sn]; pairs (2i, 2i+1) come from one register
template <int N, int S>
__device__ inline void scatter_row(int16_t* __restrict__ buf, int sn, const uint32_t (&v)[N / 2]) {
#pragma unroll
  for (int i = 0; i < N / 2; i++) {
    buf[(2 * i) * S + sn] = (int16_t)(v[i] & 0xffffu);
    buf[(2 * i + 1) * S + sn] = (int16_t)(v[i] >> 16);
  }
}
template <int N, int S>
__device__ inline void gather_row(const int16_t* __restrict__ buf, int n, uint32_t (&v)[N / 2]) {
  if constexpr (N == 4) {
    const u32x2 a = *reinterpret_cast<const u32x2*>(buf + n * S);
    v[0] = a.x; v[1] = a.y;
  } else {
#pragma unroll
    for (int i = 0; i < N / 8; i++) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(buf + n * S + 8 * i);
      v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
    }
  }
}
__device__ inline int half_of(uint32_t w, int hi) { return hi ? ((int)w >> 16) : (int)(int16_t)(w & 0xffffu); }

// Both stages for the TU of this N-lane group.  lv: row n of the TU's levels (two per register).  buf: the group's LDS tile.
// Returns row n of the residual, two samples per register.  skip: transform-skip TU (xITransformSkip, TComTrQuant.cpp:1920-1959).
template <int LOG2N>
__device__ inline void itx_tu_pk(const uint32_t (&lv)[(1 << LOG2N) / 2], int n, int per, int rem, bool skip, int bd,
                                 int16_t* __restrict__ buf, uint32_t (&res)[(1 << LOG2N) / 2]) {
  constexpr int N = 1 << LOG2N, S = PkCfg<LOG2N>::S;
  // ---- xDeQuant, flat scaling (TComTrQuant.cpp:1276-1311), on the row this lane loaded
  const int tshift = 15 - bd - LOG2N;                     // getTransformShift
  const int rshift = 6 - (tshift + per);                  // IQUANT_SHIFT - (transformShift + per)
  const int scale = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
  uint32_t d[N / 2];
  if (rshift <= 0) {
    const uint32_t sp = (uint32_t)(scale << (-rshift)) * 0x10001u;
#pragma unroll
    for (int i = 0; i < N / 2; i++) d[i] = pk_mad_sat(lv[i], sp);
  } else {
    const int add = 1 << (rshift - 1);
#pragma unroll
    for (int i = 0; i < N / 2; i++)
      d[i] = cvt_pk_sat((__mul24(half_of(lv[i], 0), scale) + add) >> rshift, (__mul24(half_of(lv[i], 1), scale) + add) >> rshift);
  }
  const int sn = slot_of(N, n);
  scatter_row<N, S>(buf, sn, d);                          // coefficient [n][x] -> tile [x][slot(n)]
  wave_lds_sync();
  uint32_t p[N / 2];
  gather_row<N, S>(buf, n, p);                            // column n of the coefficients, as pairs in slot order
  wave_lds_sync();
  uint32_t t1[N / 2];
  if (skip) {
    // stage 1 does the rounding shift, element by element; natural order out (the second hand-off un-permutes, see below)
    int o[N];
#pragma unroll
    for (int sidx = 0; sidx < N; sidx++) {
      const int c = half_of(p[sidx / 2], sidx & 1);
      o[slot_k(N, sidx)] = tshift > 0 ? (c + (1 << (tshift - 1))) >> tshift : (tshift == 0 ? c : c << (-tshift));
    }
#pragma unroll
    for (int i = 0; i < N / 2; i++) t1[i] = cvt_pk_sat(o[2 * i], o[2 * i + 1]);
  } else {
    int o[N];
    idct_pk<N, 0, N / 2>(p, 64, o);                       // shift_1st = 7 with its rounding in the seed, clip to 16 bit in the pack
#pragma unroll
    for (int i = 0; i < N / 2; i++) t1[i] = cvt_pk_sat(o[2 * i] >> 7, o[2 * i + 1] >> 7);
  }
  scatter_row<N, S>(buf, sn, t1);                         // intermediate [n][m] -> tile [m][slot(n)]
  wave_lds_sync();
  uint32_t q[N / 2];
  gather_row<N, S>(buf, n, q);                            // column n of the intermediate = what produces row n of the residual
  wave_lds_sync();                                        // the tile is reused by the next TU of this group
  if (skip) {
    // pass-through: residual [n][x] = intermediate [x][n], which sits at slot(x)
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
      constexpr int dummy = 0; (void)dummy;
      const int s0 = slot_of_c(N, 2 * i), s1 = slot_of_c(N, 2 * i + 1);
      res[i] = __builtin_amdgcn_perm(q[s1 / 2], q[s0 / 2], ((s1 & 1) ? 0x07060000u : 0x05040000u) | ((s0 & 1) ? 0x0302u : 0x0100u));
    }
  } else {
    const int shift2 = 20 - bd;                           // TRANSFORM_MATRIX_SHIFT + maxTrDynamicRange - 1 - bitDepth
    int r[N];
    idct_pk<N, 0, N / 2>(q, 1 << (shift2 - 1), r);
#pragma unroll
    for (int i = 0; i < N / 2; i++) res[i] = cvt_pk_sat(r[2 * i] >> shift2, r[2 * i + 1] >> shift2);
  }
}

}  // namespace

template <int LOG2N> struct ItxLds {
  static constexpr int BYTES = PkCfg<LOG2N>::TPB * PkCfg<LOG2N>::TU_ELEMS * 2;
};
constexpr int kItxLdsBytes = std::max(std::max(ItxLds<2>::BYTES, ItxLds<3>::BYTES), std::max(ItxLds<4>::BYTES, ItxLds<5>::BYTES));

// all coded TUs of one size class and shard; blocks stride over a list whose length only the device knows
template <int LOG2N>
__device__ inline void itx_class(const PicDev& P, int shard, int bx, int nbx, char* __restrict__ lds_raw) {
  constexpr int N = 1 << LOG2N;
  const int cls = LOG2N - 2;
  const uint32_t count = min(ldg(P.tu_count + cls * kTuShards + shard), P.tu_cap[cls]);
  const TuRec* __restrict__ list = P.tu[cls] + (size_t)shard * P.tu_cap[cls];
  const int j = threadIdx.x / N, n = threadIdx.x % N;
  int16_t* buf = reinterpret_cast<int16_t*>(lds_raw) + j * PkCfg<LOG2N>::TU_ELEMS;
  // Waves run independently.  One loop iteration of a wave covers a group of U*TPW consecutive TUs (lane slot jw takes
  // TUs base + u*TPW + jw, u < U, one after the other through the same LDS tile), and the loop is a software pipeline:
  // while group g is transformed the levels and prediction rows of group g+1 are in flight and the records of group g+2
  // are being fetched.  Measured at 2160p: U > 1 does not pay (the kernel is bound by VALU issue, not by bytes in flight).
  constexpr int TPW = 64 / N;
  constexpr int U = 1;
  const int jw = (threadIdx.x & 63) / N;                          // TU slot inside the wave
  const uint32_t stride = (uint32_t)nbx * 4 * TPW * U;
  uint32_t t = ((uint32_t)bx * 4 + (threadIdx.x >> 6)) * TPW * U + jw;
  struct Rec { uint32_t w0, w1, w2; };
  auto load_rec = [&](uint32_t ti, Rec& r) {
    r.w0 = r.w1 = r.w2 = 0;
    if (ti < count) { const uint32_t* rp = reinterpret_cast<const uint32_t*>(list + ti); r.w0 = ldg(rp); r.w1 = ldg(rp + 1); r.w2 = ldg(rp + 2); }
  };
  auto row_ptr = [&](const Rec& r) {
    const int comp = r.w1 & 3, cs = comp ? 1 : 0;
    return P.rec[comp] + (size_t)((((int)(r.w0 >> 16) * 4) >> cs) + n) * P.pitch[comp] + (((int)(r.w0 & 0xffff) * 4) >> cs);
  };
  auto load_data = [&](uint32_t ti, const Rec& r, uint32_t (&lv)[N / 2], uint32_t (&pw)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; i++) { lv[i] = 0; pw[i] = 0; }
    if (ti < count) {
      const int16_t* lev = P.coef[r.w1 & 3] + r.w2 + n * N;       // lane n: row n of the level block
      const int16_t* row = row_ptr(r);                            // lane n: row n of the prediction
      if constexpr (N == 4) {
        const u32x2 a = ldg2(lev), c = ldg2(row);
        lv[0] = a.x; lv[1] = a.y; pw[0] = c.x; pw[1] = c.y;
      } else {
#pragma unroll
        for (int i = 0; i < N / 8; i++) { const u32x4 a = ldg4(lev + i * 8); lv[4 * i] = a.x; lv[4 * i + 1] = a.y; lv[4 * i + 2] = a.z; lv[4 * i + 3] = a.w; }
#pragma unroll
        for (int i = 0; i < N / 8; i++) { const u32x4 c = ldg4(row + i * 8); pw[4 * i] = c.x; pw[4 * i + 1] = c.y; pw[4 * i + 2] = c.z; pw[4 * i + 3] = c.w; }
      }
    }
  };
  Rec rc[U], rn[U];
  uint32_t lv_c[U][N / 2], pw_c[U][N / 2];
#pragma unroll
  for (int u = 0; u < U; u++) load_rec(t + u * TPW, rc[u]);
#pragma unroll
  for (int u = 0; u < U; u++) load_rec(t + stride + u * TPW, rn[u]);
#pragma unroll
  for (int u = 0; u < U; u++) load_data(t + u * TPW, rc[u], lv_c[u], pw_c[u]);
  for (uint32_t tw = t - jw; tw < count; tw += stride, t += stride) {   // tw: wave-uniform loop variable
    uint32_t lv_n[U][N / 2], pw_n[U][N / 2];
    Rec rnn[U];
#pragma unroll
    for (int u = 0; u < U; u++) load_data(t + stride + u * TPW, rn[u], lv_n[u], pw_n[u]);
#pragma unroll
    for (int u = 0; u < U; u++) load_rec(t + 2 * stride + u * TPW, rnn[u]);
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (tw + u * TPW >= count) break;                           // wave-uniform: the rest of the group is past the end
      const bool active = t + u * TPW < count;
      const Rec r = rc[u];
      const int comp = r.w1 & 3, flags = (r.w1 & 0xff) >> 2;
      const int per = (int)(int8_t)((r.w1 >> 8) & 0xff), rem = (int)(int8_t)((r.w1 >> 16) & 0xff);
      const int bd = P.bd[comp];
      uint32_t res[N / 2];
      itx_tu_pk<LOG2N>(lv_c[u], n, per, rem, (flags & 2) != 0, bd, buf, res);
      if (active) {
        // recon row n: ClipBD(pred + resid) in place, two samples per lane operation
        const uint32_t maxv2 = (uint32_t)((1 << bd) - 1) * 0x10001u;
        int16_t* row = row_ptr(r);
        uint32_t o[N / 2];
#pragma unroll
        for (int i = 0; i < N / 2; i++) o[i] = pk_clip_u(pk_add_sat(pw_c[u][i], res[i]), maxv2);
        if constexpr (N == 4) { u32x2 v = {o[0], o[1]}; stg2(row, v); }
        else {
#pragma unroll
          for (int seg = 0; seg < N / 8; seg++) { u32x4 v = {o[seg * 4], o[seg * 4 + 1], o[seg * 4 + 2], o[seg * 4 + 3]}; stg4(row + seg * 8, v); }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      rc[u] = rn[u]; rn[u] = rnn[u];
#pragma unroll
      for (int i = 0; i < N / 2; i++) { lv_c[u][i] = lv_n[u][i]; pw_c[u][i] = pw_n[u][i]; }
    }
  }
}

// one launch for all four size classes: blockIdx.y = class * kTuShards + shard, so short lists of one class share the
// chip with the long lists of another instead of each class paying its own latency-bound launch
__global__ void __launch_bounds__(256) k_itx(const PicDev* __restrict__ pics, Batch b, uint32_t class_mask) {
  __shared__ __attribute__((aligned(16))) char lds[kItxLdsBytes];
  const PicDev& P = pics[b.pic[blockIdx.z]];
  const int cls = blockIdx.y / kTuShards, shard = blockIdx.y % kTuShards;
  if (!((class_mask >> cls) & 1)) return;                         // tuning aid: time one size class alone
  switch (cls) {
    case 0: itx_class<2>(P, shard, blockIdx.x, gridDim.x, lds); break;
    case 1: itx_class<3>(P, shard, blockIdx.x, gridDim.x, lds); break;
    case 2: itx_class<4>(P, shard, blockIdx.x, gridDim.x, lds); break;
    default: itx_class<5>(P, shard, blockIdx.x, gridDim.x, lds); break;
  }
}

void launch_itx(const PicDev* pics, const Batch& b, int log2size, uint32_t blocks_per_shard, hipStream_t s) {
  dim3 grid(blocks_per_shard, 4 * kTuShards, (unsigned)b.n);
  hipLaunchKernelGGL(k_itx, grid, dim3(256), 0, s, pics, b, log2size ? (uint32_t)log2size : 0xfu);
}

// ---- kernel-level seam: residual of n TUs from flat arrays (tests; hmgpu_inverse_transform_batch) ----------------------
template <int LOG2N>
__global__ void __launch_bounds__(256) k_itx_flat(int bit_depth, int n_tus, const int16_t* __restrict__ levels,
                                                  const int8_t* __restrict__ per, const int8_t* __restrict__ rem,
                                                  const uint8_t* __restrict__ flags, int16_t* __restrict__ resid) {
  constexpr int N = 1 << LOG2N, S = ItxCfg<LOG2N>::STRIDE, TPB = ItxCfg<LOG2N>::TPB;
  __shared__ __attribute__((aligned(16))) int lds[TPB * N * S];
  __shared__ __attribute__((aligned(16))) int16_t clds[TPB * N * N];
  const int j = threadIdx.x / N, n = threadIdx.x % N;
  int* tile = lds + j * N * S;
  int16_t* ctile = clds + j * N * N;
  for (int base = blockIdx.x * TPB; base < n_tus; base += gridDim.x * TPB) {
    const int t = base + j;
    const bool active = t < n_tus;
    uint32_t lv[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; i++) lv[i] = active ? ldg(reinterpret_cast<const uint32_t*>(levels + (size_t)t * N * N + n * N) + i) : 0u;
    const int fl = active ? flags[t] : 0;
    int16_t* row = resid + (size_t)(active ? t : 0) * N * N + n * N;
    if (__builtin_amdgcn_ballot_w64((fl & 1) != 0) != 0) {
      // a DST TU (4x4 intra luma) in this wave: the 32-bit pipeline of itx_core.h, the one k_intra uses
      int res[N];
      itx_tu<LOG2N>(lv, n, active ? per[t] : 0, active ? rem[t] : 0, fl, bit_depth, tile, ctile, res);
      if (!active) continue;
#pragma unroll
      for (int x = 0; x < N; x++) row[x] = (int16_t)res[x];
    } else {
      // the packed pipeline k_itx uses
      uint32_t res[N / 2];
      static_assert(PkCfg<LOG2N>::TPB * PkCfg<LOG2N>::TU_ELEMS * 2 <= TPB * N * S * 4, "the 32-bit tile is large enough for the 16-bit one");
      itx_tu_pk<LOG2N>(lv, n, active ? per[t] : 0, active ? rem[t] : 0, (fl & 2) != 0, bit_depth,
                       reinterpret_cast<int16_t*>(lds) + j * PkCfg<LOG2N>::TU_ELEMS, res);
      if (!active) continue;
#pragma unroll
      for (int i = 0; i < N / 2; i++) reinterpret_cast<uint32_t*>(row)[i] = res[i];
    }
  }
}

void launch_itx_flat(int log2size, int bit_depth, int n, const int16_t* levels, const int8_t* per, const int8_t* rem,
                     const uint8_t* flags, int16_t* resid, hipStream_t s) {
  const int tpb = 256 >> log2size;
  dim3 grid((unsigned)std::min(4096, (n + tpb - 1) / tpb));
  switch (log2size) {
    case 2: hipLaunchKernelGGL(k_itx_flat<2>, grid, dim3(256), 0, s, bit_depth, n, levels, per, rem, flags, resid); break;
    case 3: hipLaunchKernelGGL(k_itx_flat<3>, grid, dim3(256), 0, s, bit_depth, n, levels, per, rem, flags, resid); break;
    case 4: hipLaunchKernelGGL(k_itx_flat<4>, grid, dim3(256), 0, s, bit_depth, n, levels, per, rem, flags, resid); break;
    case 5: hipLaunchKernelGGL(k_itx_flat<5>, grid, dim3(256), 0, s, bit_depth, n, levels, per, rem, flags, resid); break;
    default: break;
  }
}

}  // namespace hmgpu
