// k_dbk.hip -- deblocking (TComLoopFilter::loopFilterPic, TComLoopFilter.cpp:130-155): one launch for all vertical
// edges of the picture, one for all horizontal edges (the second consumes the first's output, as in HM).
//
// One thread per 4-sample edge unit on the 8x8 grid, the unit at which HM takes every decision (xEdgeFilterLuma,
// :540-653).  The thread takes the unit's boundary strength, mean QP and exemptions from its EdgeRec entry (k_prep.hip:
// xGetBoundaryStrengthSingle, :411-537 -- reference pictures are compared by device picture handle, the counterpart of
// HM's TComPic* comparison), evaluates dE / side / strong-weak on lines 0 and 3 and filters its four lines in registers.  Where the unit also lies on the 8x8 chroma grid and Bs == 2 the same thread filters the two
// chroma lines of Cb and Cr (xEdgeFilterChroma, :656-785).  Filtering is in place: units on the 8x8 grid read and
// write disjoint samples (4 read / 3 written per side), so no two threads of one launch touch the same sample.
#include "hmgpu_dev.h"
#include "filter_core.h"

namespace hmgpu {

template <int DIR>
__global__ void __launch_bounds__(256) k_deblock(const PicDev* __restrict__ pics, Batch b) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  // DIR 0: thread = (edge column ex, 4-row unit uy); DIR 1: thread = (4-column unit ux, edge row ey).  lanes run along x.
  const int tx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ty = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int gx = DIR == 0 ? tx * 2 : tx;             // BlkInfo coordinates of Q
  const int gy = DIR == 0 ? ty : ty * 2;
  const int x = gx * 4, y = gy * 4;
  if (x >= P.width || y >= P.height) return;
  // the unit's EdgeRec entry (k_prep): Bs, mean QP, exemptions of the two blocks that face each other
  const uint32_t rec = ldg(reinterpret_cast<const uint16_t*>(P.edges + (size_t)(y >> 3) * (P.grid_w >> 1) + (x >> 3)) + (DIR == 0 ? ((y >> 2) & 1) : 2 + ((x >> 2) & 1)));
  const int bs = rec & 3;
  if (bs == 0) return;
  const int sidx = P.slice_idx ? ldg(P.slice_idx + (size_t)(y >> P.log2ctu) * P.ctus_w + (x >> P.log2ctu)) : 0;
  const SliceDev* slp = P.slices + sidx;                            // offsets come from the Q side's slice (:565-566)
  const int tc_off = ldg(&slp->tc_offset_div2), beta_off = ldg(&slp->beta_offset_div2);
  const int qp = (int)((rec >> 2) & 127) - 32;
  const int pitch = P.pitch[0];
  int16_t* base = P.rec[0] + (size_t)y * pitch + x;
  // the unit as line pairs (filter_core.h): la = lines 0|1, lb = lines 2|3, index = position across the edge
  uint32_t la[8], lb[8];
  if (DIR == 0) {
    uint32_t r[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x4 v = ldg4_a8(base + (size_t)i * pitch - 4);
      r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
    }
    rows_to_pairs(r[0], r[1], la);
    rows_to_pairs(r[2], r[3], lb);
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32x2 v = ldg2(base + (ptrdiff_t)(k - 4) * pitch);      // four contiguous samples = the four lines at position k
      la[k] = v.x; lb[k] = v.y;
    }
  }
  const bool p_nf = (rec >> 9) & 1, q_nf = (rec >> 10) & 1;
  filter_luma_unit(la, lb, bs, qp, tc_off, beta_off, P.bd[0], p_nf, q_nf);
  if (DIR == 0) {
    uint32_t r[4][4];
    pairs_to_rows(la, r[0], r[1]);
    pairs_to_rows(lb, r[2], r[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u32x4 o = {r[i][0], r[i][1], r[i][2], r[i][3]};
      stg4_a8(base + (size_t)i * pitch - 4, o);
    }
  } else {
#pragma unroll
    for (int k = 1; k < 7; k++) {
      u32x2 o = {la[k], lb[k]};
      stg2(base + (ptrdiff_t)(k - 4) * pitch, o);
    }
  }
  // chroma: Bs 2 only, edges on the 8-sample chroma grid = 16-sample luma grid (:225-229, :684-692, :727)
  // (4:2:2 / 4:4:4 pictures: k_deblock_chroma_fmt, k_cfmt.hip)
  if (bs == 2 && P.fmt == 1 && ((DIR == 0 ? x : y) & 15) == 0) {
    const int cp = P.pitch[1];
    const int maxc = (1 << P.bd[1]) - 1;
#pragma unroll
    for (int comp = 1; comp < 3; comp++) {
      const int tc = chroma_tc(qp, ldg(comp == 1 ? &slp->pps_cb_qp_offset : &slp->pps_cr_qp_offset), tc_off, P.bd[comp]);
      int16_t* cb = P.rec[comp] + (size_t)(y >> 1) * cp + kCStep * (x >> 1);      // (a component's samples lie kCStep elements apart)
#pragma unroll
      for (int i = 0; i < 2; i++) {
        // xPelFilterChroma (:870-891): two lines per unit
        int16_t* s = DIR == 0 ? cb + (size_t)i * cp : cb + kCStep * i;
        const ptrdiff_t o = DIR == 0 ? kCStep : cp;
        const int m2 = (uint16_t)ldg(s - 2 * o), m3 = (uint16_t)ldg(s - o), m4 = (uint16_t)ldg(s), m5 = (uint16_t)ldg(s + o);
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        if (!p_nf) stg(s - o, (int16_t)clip3(0, maxc, m3 + delta));          // xPelFilterChroma :883-890
        if (!q_nf) stg(s, (int16_t)clip3(0, maxc, m4 - delta));
      }
    }
  }
}

void launch_deblock(const PicDev* pics, const Batch& b, int dir, int width, int height, hipStream_t s) {
  if (dir == 0) {
    dim3 grid((unsigned)((width / 8 + 63) / 64), (unsigned)((height / 4 + 3) / 4), (unsigned)b.n);
    hipLaunchKernelGGL(k_deblock<0>, grid, dim3(256), 0, s, pics, b);
  } else {
    dim3 grid((unsigned)((width / 4 + 63) / 64), (unsigned)((height / 8 + 3) / 4), (unsigned)b.n);
    hipLaunchKernelGGL(k_deblock<1>, grid, dim3(256), 0, s, pics, b);
  }
}

}  // namespace hmgpu
