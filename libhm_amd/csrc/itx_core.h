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

// N-point inverse DCT, unrounded: out[k] = sum_m T_N[m][k] * in[m]; partialButterflyInverseN's even/odd structure
// (TComTrQuant.cpp:468-828).  Inputs are 17-bit signed, so v_mad_i32_i24 is exact.
template <int N>
__device__ inline void idct_1d(const int (&in)[N], int (&out)[N]) {
  if constexpr (N == 2) {
    out[0] = __mul24(64, in[0]) + __mul24(64, in[1]);
    out[1] = __mul24(64, in[0]) - __mul24(64, in[1]);
  } else {
    int ev[N / 2], e[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; i++) ev[i] = in[2 * i];
    idct_1d<N / 2>(ev, e);
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
      int o = 0;
#pragma unroll
      for (int m = 1; m < N; m += 2) o += __mul24(tmat(N, m, k), in[m]);
      out[k] = e[k] + o;
      out[N - 1 - k] = e[k] - o;
    }
  }
}

__device__ inline void idst_4(const int (&in)[4], int (&out)[4]) {     // fastInverseDst, TComTrQuant.cpp:437
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int s = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) s += __mul24(dst4(m, k), in[m]);
    out[k] = s;
  }
}

// LDS hand-off between the lanes of ONE wave (a TU never spans waves): DS instructions of a wave execute in order, so all
// that is needed is to stop the compiler from moving LDS accesses across this point and to wait for outstanding DS writes.
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LOG2N> struct ItxCfg {
  static constexpr int N = 1 << LOG2N;
  static constexpr int STRIDE = N == 4 ? 4 : N + 4;       // dwords per LDS row (conflict-free ds_read_b128)
  static constexpr int TPB = 256 / N;                     // TUs per 256-thread block
};

// Both stages for the TU of this N-lane group.  lev: the TU's N*N levels (row-major, contiguous).  tile: the group's
// 32-bit LDS tile for the intermediates, ctile: its 16-bit tile for the levels.  Returns row `n` of the residual in res[].
// flags: bit0 DST, bit1 transform skip.
template <int LOG2N>
__device__ inline void itx_tu(const uint32_t (&lv)[(1 << LOG2N) / 2], int n, int per, int rem, int flags, int bd,
                              int* __restrict__ tile, int16_t* __restrict__ ctile, int (&res)[1 << LOG2N]) {
  constexpr int N = 1 << LOG2N, S = ItxCfg<LOG2N>::STRIDE;
  // ---- the level block came in with 16-byte loads (lane n holds row n) and is re-read by columns from LDS
  if constexpr (N == 4) {
    u32x2 v = {lv[0], lv[1]};
    *reinterpret_cast<u32x2*>(ctile + n * 4) = v;
  } else {
#pragma unroll
    for (int i = 0; i < N / 8; i++) {
      u32x4 v = {lv[4 * i], lv[4 * i + 1], lv[4 * i + 2], lv[4 * i + 3]};
      *reinterpret_cast<u32x4*>(ctile + n * N + i * 8) = v;
    }
  }
  wave_lds_sync();
  // ---- xDeQuant, flat scaling (TComTrQuant.cpp:1276-1311) on column n
  const int tshift = 15 - bd - LOG2N;                     // getTransformShift
  const int rshift = 6 - (tshift + per);                  // IQUANT_SHIFT - (transformShift + per)
  const int scale = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
  int c[N];
#pragma unroll
  for (int m = 0; m < N; m++) {
    const int q = (int)ctile[m * N + n];
    int v;
    if (rshift > 0) v = (__mul24(q, scale) + (1 << (rshift - 1))) >> rshift;
    else v = (int)((unsigned)__mul24(q, scale) << (-rshift));
    c[m] = clip3(-32768, 32767, v);
  }
  int o[N];
  if (flags & 2) {
    // xITransformSkip (TComTrQuant.cpp:1920-1959): stage 1 does the rounding shift, stage 2 passes the row through
#pragma unroll
    for (int m = 0; m < N; m++) o[m] = tshift > 0 ? (c[m] + (1 << (tshift - 1))) >> tshift : (tshift == 0 ? c[m] : c[m] << (-tshift));
  } else {
    if (LOG2N == 2 && (flags & 1)) { int t4[4] = {c[0], c[1], c[2], c[3]}, r4[4]; idst_4(t4, r4);
#pragma unroll
      for (int k = 0; k < 4; k++) o[k] = r4[k]; }
    else idct_1d<N>(c, o);
#pragma unroll
    for (int k = 0; k < N; k++) o[k] = clip3(-32768, 32767, (o[k] + 64) >> 7);          // shift_1st = 7, clip to 16 bit
  }
#pragma unroll
  for (int k = 0; k < N; k++) tile[k * S + n] = o[k];     // T1[row k][column n]
  wave_lds_sync();
  int r[N];
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    const int4 v = *reinterpret_cast<const int4*>(&tile[n * S + i]);
    r[i] = v.x; r[i + 1] = v.y; r[i + 2] = v.z; r[i + 3] = v.w;
  }
  wave_lds_sync();                                         // tile is reused by the next TU of this group
  if (flags & 2) {
#pragma unroll
    for (int x = 0; x < N; x++) res[x] = (int)(int16_t)r[x];
  } else {
    const int shift2 = 20 - bd;                            // TRANSFORM_MATRIX_SHIFT + maxTrDynamicRange - 1 - bitDepth
    if (LOG2N == 2 && (flags & 1)) { int t4[4] = {r[0], r[1], r[2], r[3]}, r4[4]; idst_4(t4, r4);
#pragma unroll
      for (int k = 0; k < 4; k++) res[k] = r4[k]; }
    else idct_1d<N>(r, res);
#pragma unroll
    for (int x = 0; x < N; x++) res[x] = clip3(-32768, 32767, (res[x] + (1 << (shift2 - 1))) >> shift2);
  }
}

}  // namespace hmgpu
