// itx_core.h -- de-quantisation + 2-D inverse transform of ONE transform unit by the N lanes that own it (shared by the
// inter residual kernel k_itx.hip and the intra reconstruction kernel k_intra.hip).
//   TComTrQuant::invTransformNxN -> xDeQuant (flat) -> xIT -> xITrMxN / xITransformSkip   TComTrQuant.cpp:1423,1203,1836,894,1920
#pragma once
#include "hmgpu_dev.h"

namespace hmgpu {

// HM g_aiT<N>[TRANSFORM_INVERSE][k][n] (TComRom.cpp:335-417): the 32-point basis sampled at odd multiples of pi/64.
__host__ __device__ constexpr int cos64(int a) {
  constexpr int t[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                         61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
  return t[a];
}
__host__ __device__ constexpr int tmat(int n_size, int k, int n) {
  const int a = ((2 * n + 1) * k * (32 / n_size)) & 127;
  return a <= 32 ? cos64(a) : (a <= 64 ? -cos64(64 - a) : (a <= 96 ? -cos64(a - 64) : cos64(128 - a)));
}
__host__ __device__ constexpr int dst4(int m, int k) {       // g_as_DST_MAT_4[TRANSFORM_INVERSE] (TComRom.cpp:456-484)
  constexpr int t[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
  return t[m][k];
}

// LDS hand-off between the lanes of ONE wave (a TU never spans waves): DS instructions of a wave execute in order, so all
// that is needed is to stop the compiler from moving LDS accesses across this point and to wait for outstanding DS writes.
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =====================================================================================================================
// Packed 16-bit formulation of the TU pipeline.
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
// Returns row n of the residual, two samples per register.  skip: transform-skip TU (xITransformSkip, TComTrQuant.cpp:1920-1959);
// dst: 4x4 intra luma TU, fastInverseDst (TComTrQuant.cpp:437-466) instead of the DCT.
template <int N>
__device__ inline void idst4_pk(const uint32_t (&p)[N / 2], int seed, int (&out)[N]) {
  if constexpr (N == 4) {
    // slot order of a 4-point column: pairs (in[1], in[3]) and (in[2], in[0])
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = dot2c(p[1], pkc(dst4(2, k), dst4(0, k)), dot2_v(p[0], pkc(dst4(1, k), dst4(3, k)), seed));
  }
}
// mrow: row n of the TU's scaling-list matrix (one byte per position, already replicated / DC-patched for 16x16 and 32x32:
// TComTrQuant.cpp:2992-3012, 3092-3106), or nullptr for flat scaling.
template <int LOG2N>
__device__ inline void itx_tu_pk(const uint32_t (&lv)[(1 << LOG2N) / 2], int n, int per, int rem, bool skip, int bd,
                                 int16_t* __restrict__ buf, uint32_t (&res)[(1 << LOG2N) / 2], bool dst = false,
                                 const uint8_t* __restrict__ mrow = nullptr, bool bypass = false) {
  constexpr int N = 1 << LOG2N, S = PkCfg<LOG2N>::S;
  if (bypass) {
    // cu_transquant_bypass: the residual IS the level block (invTransformNxN, TComTrQuant.cpp:1440-1470).  All N lanes of the TU
    // take this exit together; the hand-offs below are wave-local orderings, not barriers, so other TUs of the wave go on.
#pragma unroll
    for (int i = 0; i < N / 2; i++) res[i] = lv[i];
    return;
  }
  // ---- xDeQuant, flat scaling (TComTrQuant.cpp:1276-1311), on the row this lane loaded
  const int tshift = 15 - bd - LOG2N;                     // getTransformShift
  const int rshift = 6 - (tshift + per);                  // IQUANT_SHIFT - (transformShift + per)
  const int scale = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
  uint32_t d[N / 2];
  if (mrow != nullptr) {
    // scaling lists (TComTrQuant.cpp:1238-1275): factor scale * m per position, four more fractional bits, 32-bit products;
    // the level is first clipped to what keeps level * factor inside 32 bits (only binding for rshift < -1)
    const int rs = rshift + 4;
    const int tb = min(16, 17 + rs), in_max = (1 << (tb - 1)) - 1, in_min = -in_max - 1;
    uint32_t mw[N / 4];
    if constexpr (N == 4) mw[0] = ldg(reinterpret_cast<const uint32_t*>(mrow));
    else {
#pragma unroll
      for (int i = 0; i < N / 16; i++) { const u32x4 a = ldg4(mrow + 16 * i); mw[4 * i] = a.x; mw[4 * i + 1] = a.y; mw[4 * i + 2] = a.z; mw[4 * i + 3] = a.w; }
      if constexpr (N == 8) { const u32x2 a = ldg2(mrow); mw[0] = a.x; mw[1] = a.y; }
    }
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
      const int m0 = (mw[i / 2] >> (16 * (i & 1))) & 0xff, m1 = (mw[i / 2] >> (16 * (i & 1) + 8)) & 0xff;
      const int q0 = clip3(in_min, in_max, half_of(lv[i], 0)), q1 = clip3(in_min, in_max, half_of(lv[i], 1));
      int c0 = __mul24(q0, scale * m0), c1 = __mul24(q1, scale * m1);
      if (rs > 0) { c0 = (c0 + (1 << (rs - 1))) >> rs; c1 = (c1 + (1 << (rs - 1))) >> rs; }
      else { c0 = (int)((unsigned)c0 << (-rs)); c1 = (int)((unsigned)c1 << (-rs)); }
      d[i] = cvt_pk_sat(c0, c1);
    }
  } else if (rshift <= 0) {
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
    if (LOG2N == 2 && dst) idst4_pk<N>(p, 64, o);
    else idct_pk<N, 0, N / 2>(p, 64, o);                  // shift_1st = 7 with its rounding in the seed, clip to 16 bit in the pack
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
    if (LOG2N == 2 && dst) idst4_pk<N>(q, 1 << (shift2 - 1), r);
    else idct_pk<N, 0, N / 2>(q, 1 << (shift2 - 1), r);
#pragma unroll
    for (int i = 0; i < N / 2; i++) res[i] = cvt_pk_sat(r[2 * i] >> shift2, r[2 * i + 1] >> shift2);
  }
}

// sps_range_extension(): what follows a block that skipped the transform (transform-skip or cu_transquant_bypass).  res: row n of
// the N x N residual; the N lanes of the TU call together.  rotate: the 4x4 block read back to front (TComTrQuant.cpp:1475-1487,
// :1943; TComTU::isNonTransformedResidualRotated).  rdpcm: invRdpcmNxN (:1737-1792), 1 = running sums along the row, 2 = down the
// columns; the sums wrap at 16 bits like HM's Pel buffer.
template <int LOG2N>
__device__ inline void resid_rotate_rdpcm(uint32_t (&res)[(1 << LOG2N) / 2], int n, bool rotate, int rdpcm) {
  constexpr int N = 1 << LOG2N;
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  if constexpr (N == 4) {
    if (rotate) {
      const uint32_t a = (uint32_t)__shfl_xor((int)res[1], 3), b = (uint32_t)__shfl_xor((int)res[0], 3);     // row 3 - n, then mirrored
      res[0] = (a >> 16) | (a << 16); res[1] = (b >> 16) | (b << 16);
    }
  }
  if (rdpcm == 1) {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
      const uint32_t lo = (res[i] + carry) & 0xffffu, hi = ((res[i] >> 16) + lo) & 0xffffu;
      res[i] = lo | (hi << 16); carry = hi;
    }
  } else if (rdpcm == 2) {
#pragma unroll
    for (int d = 1; d < N; d <<= 1) {
#pragma unroll
      for (int i = 0; i < N / 2; i++) {
        const uint32_t up = (uint32_t)__shfl_up((int)res[i], d, N);           // row n - d of the same TU
        const u16x2 s = __builtin_bit_cast(u16x2, res[i]) + __builtin_bit_cast(u16x2, up);
        if (n >= d) res[i] = __builtin_bit_cast(uint32_t, s);
      }
    }
  }
}

}  // namespace hmgpu
