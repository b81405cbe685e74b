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
  // are being fetched.  Measured at 2160p: U > 1 does not pay, not even for the 4x4 / 8x8 classes alone (U = 4 / 2: +3 %
  // time) -- those classes move 1.1 / 1.9 TB/s of algorithmic bytes but 32-byte sectors for 8- and 16-byte rows.
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
      // scaling lists (inter TUs: list 3 + component; not for transform-skip blocks other than 4x4, TComTrQuant.h:180)
      const uint8_t* mrow = (P.sl_m != nullptr && (!(flags & 2) || N == 4)) ? P.sl_m + (((LOG2N - 2) * 6 + 3 + comp) << 10) + n * N : nullptr;
      itx_tu_pk<LOG2N>(lv_c[u], n, per, rem, (flags & 2) != 0, bd, buf, res, false, mrow, (flags & 4) != 0);
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
  constexpr int N = 1 << LOG2N, TPB = PkCfg<LOG2N>::TPB;
  __shared__ __attribute__((aligned(16))) int16_t lds[TPB * PkCfg<LOG2N>::TU_ELEMS];
  const int j = threadIdx.x / N, n = threadIdx.x % N;
  for (int base = blockIdx.x * TPB; base < n_tus; base += gridDim.x * TPB) {
    const int t = base + j;
    const bool active = t < n_tus;
    uint32_t lv[N / 2], res[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; i++) lv[i] = active ? ldg(reinterpret_cast<const uint32_t*>(levels + (size_t)t * N * N + n * N) + i) : 0u;
    const int fl = active ? flags[t] : 0;
    itx_tu_pk<LOG2N>(lv, n, active ? per[t] : 0, active ? rem[t] : 0, (fl & 2) != 0, bit_depth, lds + j * PkCfg<LOG2N>::TU_ELEMS, res,
                     (fl & 1) != 0);
    if (!active) continue;
    uint32_t* row = reinterpret_cast<uint32_t*>(resid + (size_t)t * N * N + n * N);
#pragma unroll
    for (int i = 0; i < N / 2; i++) row[i] = res[i];
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
