// k_cfmt.hip -- what the chroma of 4:2:2 / 4:4:4 pictures needs beyond the kernels every picture shares.
//   TComTrQuant::crossComponentPrediction (4:4:4)                        TComTrQuant.cpp:3294-3335, called from :1591-1606 and TDecCu.cpp:583-612
//   TComLoopFilter::xEdgeFilterChroma on the format's own edge grid      TComLoopFilter.cpp:656-785
// Luma of these pictures runs through the kernels of 4:2:0 pictures unchanged.  Chroma: k_prep lists the chroma blocks of every
// transform unit in the format's shapes (4:4:4: the luma blocks' twins; 4:2:2: two squares per block), k_itx transforms them, the kernels
// here add the cross-component term to the residual tiles and filter the chroma edges; motion compensation of these formats is the
// register path of k_mc_cells.hip over every inter cell, intra prediction k_intra.hip with the component's geometry, SAO k_sao.hip.
// Correct first: none of this is tuned -- BASELINE's configurations are 4:2:0.
#include "hmgpu_dev.h"
#include "filter_core.h"

namespace hmgpu {

// ---- cross-component prediction: residual_C += (alpha * (residual_Y >> (bit depth Y - bit depth C))) >> 3 over every 4x4 partition whose
// transform unit carries a weight (the weight is stored for all partitions of the unit, m_crossComponentPredictionAlpha; it is only ever
// non-zero where the unit's luma block is coded: TDecEntropy.cpp:560-563).  One thread per partition and chroma component; the luma and
// chroma tiles have the same geometry in 4:4:4.
__global__ void __launch_bounds__(256) k_ccp(const PicDev* __restrict__ pics, Batch b) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  if (P.ccp[0] == nullptr) return;
  const int comp = 1 + (int)blockIdx.y;
  const int gp = blockIdx.x * 256 + threadIdx.x;
  if (gp >= b.num_ctus[blockIdx.z] * P.parts) return;
  const int ctu = b.first_ctu[blockIdx.z] + gp / P.parts, z = gp % P.parts;
  const int alpha = (int)ldg(P.ccp[comp - 1] + (size_t)ctu * P.parts + z);
  if (alpha == 0) return;
  const int x = ((ctu % P.ctus_w) << P.log2ctu) + 4 * zscan_x(z), y0 = ((ctu / P.ctus_w) << P.log2ctu) + 4 * zscan_y(z);
  if (x >= P.width || y0 >= P.height || (int)ldg(P.part_size + (size_t)ctu * P.parts + z) == HMGPU_SIZE_NONE) return;
  const int rtw = P.grid_w >> 1, diff = P.bd[0] - P.bd[comp];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int y = y0 + r;
    const size_t o = ((size_t)((y >> 3) * rtw + (x >> 3)) * 8 + resid_slot(y)) * 8 + (x & 7);
    const u32x2 l = ldg2(P.resid[0] + o);
    u32x2 c = ldg2(P.resid[comp] + o);
    const uint32_t lw[2] = {l.x, l.y};
    uint32_t cw[2] = {c.x, c.y};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      int out[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int lv = (int)(int16_t)(lw[k] >> (16 * h)), cv = (int)(int16_t)(cw[k] >> (16 * h));
        out[h] = cv + ((alpha * (diff >= 0 ? lv >> diff : lv << -diff)) >> 3);            // (a Pel: wraps at 16 bits like HM's buffer)
      }
      cw[k] = ((uint32_t)out[0] & 0xffffu) | ((uint32_t)out[1] << 16);
    }
    stg2(P.resid[comp] + o, u32x2{cw[0], cw[1]});
  }
}

void launch_ccp(const PicDev* pics, const Batch& b, int max_ctus, hipStream_t s) {
  // (parts per CTU: at most 256)
  dim3 grid((unsigned)(((size_t)max_ctus * 256 + 255) / 256), 2, (unsigned)b.n);
  hipLaunchKernelGGL(k_ccp, grid, dim3(256), 0, s, pics, b);
}

// ---- chroma deblocking of 4:2:2 / 4:4:4 pictures.  One thread per 4-sample luma edge unit on the 8x8 luma grid, as k_deblock: the unit's
// EdgeRec entry gives Bs and the mean QP; chroma is filtered where Bs == 2 and the edge lies on the 8-sample grid of the CHROMA plane
// (uiEdgeNumInLCU % (DEBLOCK_SMALLEST_BLOCK / uiPelsInPartChroma), :684-692): every 16 luma samples across a subsampled direction, every 8
// otherwise; the unit's (4 >> cs) chroma lines along the edge.
template <int DIR>
__global__ void __launch_bounds__(256) k_deblock_chroma_fmt(const PicDev* __restrict__ pics, Batch b) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  const int tx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ty = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int x = (DIR == 0 ? tx * 2 : tx) * 4, y = (DIR == 0 ? ty : ty * 2) * 4;      // luma position of the unit's Q side
  if (x >= P.width || y >= P.height) return;
  const int across = DIR == 0 ? x >> P.csx : y >> P.csy;
  if (across & 7) return;
  const uint32_t rec = ldg(reinterpret_cast<const uint16_t*>(P.edges + (size_t)(y >> 3) * (P.grid_w >> 1) + (x >> 3)) + (DIR == 0 ? ((y >> 2) & 1) : 2 + ((x >> 2) & 1)));
  if ((rec & 3) != 2) return;
  const int sidx = P.slice_idx ? ldg(P.slice_idx + (size_t)(y >> P.log2ctu) * P.ctus_w + (x >> P.log2ctu)) : 0;
  const SliceDev* slp = P.slices + sidx;
  const int tc_off = ldg(&slp->tc_offset_div2);
  const int qp = (int)((rec >> 2) & 127) - 32;
  const bool p_nf = (rec >> 9) & 1, q_nf = (rec >> 10) & 1;
  const int cp = P.pitch[1], maxc = (1 << P.bd[1]) - 1;
  const int lines = DIR == 0 ? 4 >> P.csy : 4 >> P.csx;
#pragma unroll
  for (int comp = 1; comp < 3; comp++) {
    const int tc = chroma_tc(qp, ldg(comp == 1 ? &slp->pps_cb_qp_offset : &slp->pps_cr_qp_offset), tc_off, P.bd[comp], P.fmt);
    int16_t* cb = P.rec[comp] + (size_t)(y >> P.csy) * cp + kCStep * (x >> P.csx);
    for (int i = 0; i < lines; i++) {
      int16_t* s = DIR == 0 ? cb + (size_t)i * cp : cb + kCStep * i;
      const ptrdiff_t o = DIR == 0 ? kCStep : cp;
      const int m2 = (uint16_t)ldg(s - 2 * o), m3 = (uint16_t)ldg(s - o), m4 = (uint16_t)ldg(s), m5 = (uint16_t)ldg(s + o);
      const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
      if (!p_nf) stg(s - o, (int16_t)clip3(0, maxc, m3 + delta));          // xPelFilterChroma :883-890
      if (!q_nf) stg(s, (int16_t)clip3(0, maxc, m4 - delta));
    }
  }
}

void launch_deblock_chroma_fmt(const PicDev* pics, const Batch& b, int dir, int width, int height, hipStream_t s) {
  if (dir == 0) {
    dim3 grid((unsigned)((width / 8 + 63) / 64), (unsigned)((height / 4 + 3) / 4), (unsigned)b.n);
    hipLaunchKernelGGL(k_deblock_chroma_fmt<0>, grid, dim3(256), 0, s, pics, b);
  } else {
    dim3 grid((unsigned)((width / 4 + 63) / 64), (unsigned)((height / 8 + 3) / 4), (unsigned)b.n);
    hipLaunchKernelGGL(k_deblock_chroma_fmt<1>, grid, dim3(256), 0, s, pics, b);
  }
}

}  // namespace hmgpu
