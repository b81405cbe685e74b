// k_itx.hip -- de-quantisation + inverse transform of the coded TUs (inter CUs and, ahead of k_intra, intra CUs).
//   TComTrQuant::invTransformNxN -> xDeQuant (flat) -> xIT -> xITrMxN / xITransformSkip   TComTrQuant.cpp:1423,1203,1836,894,1920
//
// One N-lane group per N x N TU (N = 4, 8, 16, 32; a wave holds 64/N TUs).  Stage 1: lane n owns coefficient column
// n (coalesced 2-byte reads along the row), de-quantises it and runs the N-point 1-D inverse transform in registers
// with the even/odd decomposition (full-rate 24-bit integer multiply-adds, all matrix entries are immediates).  The
// 16-bit-clipped intermediates cross to the row owners through a padded LDS tile (conflict-free b32 writes / b128
// reads).  Stage 2: lane y owns row y of the residual, transforms it and writes it to the picture's residual tiles
// (PicDev::resid: 8x8 samples = one 128-byte line per tile), where the motion-compensation kernels pick it up when they write the
// prediction: reconstruction = ClipBD(prediction + residual) happens there (TComYuv::addClip, TComYuv.cpp:264), the prediction never
// makes the round trip through HBM and this kernel is a pure stream -- levels in, residual out, every request a full line (rows of
// a TU scattered over the picture plane, 8 to 64 bytes each, were what held its first version at a third of the roofline).
// TUs come from the per-class lists built by k_prep; blocks stride over a list whose length only the device knows.
#include "hmgpu_dev.h"
#include "itx_core.h"
#include <algorithm>

namespace hmgpu {

template <int LOG2N> struct ItxLds {
  static constexpr int BYTES = PkCfg<LOG2N>::TPB * PkCfg<LOG2N>::TU_ELEMS * 2;
};
constexpr int kItxLdsBytes = std::max(std::max(ItxLds<2>::BYTES, ItxLds<3>::BYTES), std::max(ItxLds<4>::BYTES, ItxLds<5>::BYTES));

// what a workgroup needs of its picture, taken from kernel-argument memory (ItxArgs): no descriptor in global memory stands between
// the start of a wave and the load of its first TU records
struct ItxPic {
  const TuRec* list; uint32_t count_cap; const uint32_t* count;
  const int16_t* coef[3]; int16_t* resid[3]; int32_t rtw[3], bd[3]; const uint8_t* sl_m;
  int32_t csx, csy;                       // chroma subsampling: a record carries the LUMA position of its block's first partition
};
template <typename T> __device__ inline T by_comp(const T (&v)[3], int comp) { return comp == 0 ? v[0] : comp == 1 ? v[1] : v[2]; }   // registers, not an indexed array

// all coded TUs of one size class and shard; blocks stride over a list whose length only the device knows
template <int LOG2N>
__device__ inline void itx_class(const ItxPic& P, int bx, int nbx, char* __restrict__ lds_raw) {
  constexpr int N = 1 << LOG2N;
  const TuRec* __restrict__ list = P.list;
  const uint32_t cap = P.count_cap;
  const int j = threadIdx.x / N, n = threadIdx.x % N;
  int16_t* buf = reinterpret_cast<int16_t*>(lds_raw) + j * PkCfg<LOG2N>::TU_ELEMS;
  // Waves run independently.  One loop iteration of a wave covers TPW consecutive TUs (lane slot jw takes TU base + jw), and
  // the loop is a software pipeline: while group g is transformed the levels of group g+1 are in flight and the records of group
  // g+2 are being fetched.  The first records are fetched together with the list length (records past the length are stale but
  // inside the list's capacity; nothing is done with them).
  constexpr int TPW = 64 / N;
  const int jw = (threadIdx.x & 63) / N;                          // TU slot inside the wave
  const uint32_t stride = (uint32_t)nbx * 4 * TPW;
  uint32_t t = ((uint32_t)bx * 4 + (threadIdx.x >> 6)) * TPW + jw;
  struct Rec { uint32_t w0, w1, w2; };
  auto load_rec = [&](uint32_t ti, Rec& r) {
    r.w0 = r.w1 = r.w2 = 0;
    if (ti < cap) { const uint32_t* rp = reinterpret_cast<const uint32_t*>(list + ti); r.w0 = ldg(rp); r.w1 = ldg(rp + 1); r.w2 = ldg(rp + 2); }
  };
  Rec rc, rn;
  load_rec(t, rc);
  load_rec(t + stride, rn);
  const uint32_t count = min(ldg(P.count), cap);
  auto load_levels = [&](uint32_t ti, const Rec& r, uint32_t (&lv)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; i++) lv[i] = 0;
    if (ti < count) {
      const int16_t* lev = by_comp(P.coef, r.w1 & 3) + r.w2 + n * N;       // lane n: row n of the level block
      if constexpr (N == 4) { const u32x2 a = ldg2(lev); lv[0] = a.x; lv[1] = a.y; }
      else {
#pragma unroll
        for (int i = 0; i < N / 8; i++) { const u32x4 a = ldg4(lev + i * 8); lv[4 * i] = a.x; lv[4 * i + 1] = a.y; lv[4 * i + 2] = a.z; lv[4 * i + 3] = a.w; }
      }
    }
  };
  uint32_t lv_c[N / 2];
  load_levels(t, rc, lv_c);
  for (uint32_t tw = t - jw; tw < count; tw += stride, t += stride) {   // tw: wave-uniform loop variable
    constexpr bool AHEAD = N < 32;                            // levels of the next group in flight (32x32: see below)
    uint32_t lv_n[AHEAD ? N / 2 : 1];
    Rec rnn;
    if constexpr (AHEAD) load_levels(t + stride, rn, lv_n);
    load_rec(t + 2 * stride, rnn);
    {
      const Rec r = rc;
      const int comp = r.w1 & 3, flags = (r.w1 & 0xff) >> 2;
      const int per = (int)(int8_t)((r.w1 >> 8) & 0xff), rem = (int)(int8_t)((r.w1 >> 16) & 0xff);
      const int bd = by_comp(P.bd, comp);
      uint32_t res[N / 2];
      // scaling lists (getScalingListType: list 3 * inter + component; not for transform-skip blocks other than 4x4, TComTrQuant.h:180)
      const uint8_t* mrow = (P.sl_m != nullptr && (!(flags & 2) || N == 4)) ? P.sl_m + (((LOG2N - 2) * 6 + ((flags & 32) ? 0 : 3) + comp) << 10) + n * N : nullptr;
      itx_tu_pk<LOG2N>(lv_c, n, per, rem, (flags & 2) != 0, bd, buf, res, (flags & 1) != 0, mrow, (flags & 4) != 0);   // DST: 4x4 intra luma (TComTU::useDST)
      if ((flags & 0x18) | (r.w1 >> 24)) resid_rotate_rdpcm<LOG2N>(res, n, ((r.w1 >> 24) & 1) != 0, (flags >> 3) & 3);   // RExt: rotation (intra 4x4), RDPCM
      if (t < count) {
        // row n of the residual into the tiles it crosses: the eight lanes that hold the rows of one tile write its 128 bytes
        const int x = ((int)(r.w0 & 0xffff) * 4) >> (comp ? P.csx : 0), y = ((((int)(r.w0 >> 16) * 4) >> (comp ? P.csy : 0))) + n;
        int16_t* row = by_comp(P.resid, comp) + ((size_t)((y >> 3) * by_comp(P.rtw, comp) + (x >> 3)) * 8 + resid_slot(y)) * 8;
        if constexpr (N == 4) stg2(row + (x & 4), u32x2{res[0], res[1]});
        else {
#pragma unroll
          for (int i = 0; i < N / 8; i++) stg4(row + i * 64, u32x4{res[4 * i], res[4 * i + 1], res[4 * i + 2], res[4 * i + 3]});
        }
      }
    }
    rc = rn; rn = rnn;
    if constexpr (AHEAD) {
#pragma unroll
      for (int i = 0; i < N / 2; i++) lv_c[i] = lv_n[i];
    } else {
      load_levels(t + stride, rc, lv_c);                           // 32x32: a second set of levels in registers costs two waves per SIMD (82 -> 69 VGPRs: 5 -> 7; measured 6 % faster)
    }
  }
}

// One launch for all four size classes: short lists of one class share the chip with the long lists of another instead of each
// class paying its own latency-bound launch.  Workgroups go round-robin to the 8 XCDs by linear id and gridDim.x is a multiple of 8:
// shard s (the TUs of every eighth group of four CTUs, all size classes) is the work of XCD s alone, so a 128-byte line of a
// picture is only ever fetched into one L2.
__global__ void __launch_bounds__(256) k_itx(const ItxArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[kItxLdsBytes];
  const int cls = blockIdx.y, shard = blockIdx.x & (kTuShards - 1), z = blockIdx.z;
  const int bx = blockIdx.x / kTuShards, nbx = a.blocks[cls];
  if (bx >= nbx) return;                                          // the lists of large TUs are short: fewer, longer-lived workgroups
  ItxPic P;
  P.list = a.tu[z][cls] + (size_t)shard * a.tu_cap[cls];
  P.count_cap = a.tu_cap[cls];
  P.count = a.tu_count[z] + cls * kTuShards + shard;
#pragma unroll
  for (int k = 0; k < 3; k++) { P.coef[k] = a.coef[z][k]; P.resid[k] = a.resid[z][k]; P.rtw[k] = a.rtw[k]; P.bd[k] = a.bd[k]; }
  P.sl_m = a.sl_m[z];
  P.csx = a.csx; P.csy = a.csy;
  switch (cls) {
    case 0: itx_class<2>(P, bx, nbx, lds); break;
    case 1: itx_class<3>(P, bx, nbx, lds); break;
    case 2: itx_class<4>(P, bx, nbx, lds); break;
    default: itx_class<5>(P, bx, nbx, lds); break;
  }
}

#ifndef ITX_B0
#define ITX_B0 1
#define ITX_B1 1
#define ITX_B2 1
#define ITX_B3 1
#endif
void launch_itx(ItxArgs& a, uint32_t blocks_per_shard, hipStream_t s) {
#ifdef ITX_BPS
  blocks_per_shard = ITX_BPS;
#endif
  const uint32_t div[4] = {ITX_B0, ITX_B1, ITX_B2, ITX_B3};
  for (int k = 0; k < 4; k++) a.blocks[k] = (a.class_mask >> k) & 1 ? (int32_t)std::max(1u, blocks_per_shard / div[k]) : 0;
  dim3 grid(blocks_per_shard * kTuShards, 4, (unsigned)a.n);
  hipLaunchKernelGGL(k_itx, grid, dim3(256), 0, s, a);
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
