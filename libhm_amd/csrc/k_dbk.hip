// k_dbk.hip -- deblocking (TComLoopFilter::loopFilterPic, TComLoopFilter.cpp:130-155): one launch for all vertical
// edges of the picture, one for all horizontal edges (the second consumes the first's output, as in HM).
//
// One thread per 4-sample edge unit on the 8x8 grid, the unit at which HM takes every decision (xEdgeFilterLuma,
// :540-653).  The thread derives the boundary strength from the two BlkInfo records that face each other across the
// edge (xGetBoundaryStrengthSingle, :411-537 -- reference pictures are compared by device picture handle, the
// counterpart of HM's TComPic* comparison), evaluates dE / side / strong-weak on lines 0 and 3 and filters its four
// lines in registers.  Where the unit also lies on the 8x8 chroma grid and Bs == 2 the same thread filters the two
// chroma lines of Cb and Cr (xEdgeFilterChroma, :656-785).  Filtering is in place: units on the 8x8 grid read and
// write disjoint samples (4 read / 3 written per side), so no two threads of one launch touch the same sample.
#include "hmgpu_dev.h"

namespace hmgpu {

__constant__ uint8_t c_tc_table[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ uint8_t c_beta_table[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                         16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54,
                                         56, 58, 60, 62, 64};
__constant__ uint8_t c_chroma_scale_420_dbk[58] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};

__device__ inline bool mvd4(const int16_t* a, const int16_t* b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }

// xGetBoundaryStrengthSingle (TComLoopFilter.cpp:411-537).  `transform_edge` is the m_aapucBS marker.  The P-slice
// branch (:515-532) is the B-slice branch with list 1 absent on both sides, so one formula serves both.
__device__ inline int boundary_strength(const BlkInfo& p, const BlkInfo& q, bool transform_edge) {
  if ((p.flags | q.flags) & BF_INTRA) return 2;
  if (transform_edge && ((p.flags | q.flags) & BF_CBFY)) return 1;
  const int p0 = p.ref[0], p1 = p.ref[1], q0 = q.ref[0], q1 = q.ref[1];
  if ((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0)) {
    if (p0 != p1) {
      if (p0 == q0) return (mvd4(q.mv[0], p.mv[0]) || mvd4(q.mv[1], p.mv[1])) ? 1 : 0;
      return (mvd4(q.mv[1], p.mv[0]) || mvd4(q.mv[0], p.mv[1])) ? 1 : 0;
    }
    return ((mvd4(q.mv[0], p.mv[0]) || mvd4(q.mv[1], p.mv[1])) && (mvd4(q.mv[1], p.mv[0]) || mvd4(q.mv[0], p.mv[1]))) ? 1 : 0;
  }
  return 1;
}

// one line across a luma edge: s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3   (xPelFilterLuma, :800-859)
__device__ inline void filter_luma_line(int (&s)[8], int tc, bool strong, int thr_cut, bool filt_p, bool filt_q, int maxv) {
  const int m0 = s[0], m1 = s[1], m2 = s[2], m3 = s[3], m4 = s[4], m5 = s[5], m6 = s[6], m7 = s[7];
  if (strong) {
    s[3] = clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    s[4] = clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    s[2] = clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    s[5] = clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    s[1] = clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[6] = clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  } else {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < thr_cut) {
      const int tc2 = tc >> 1;
      delta = clip3(-tc, tc, delta);
      s[3] = clip3(0, maxv, m3 + delta);
      s[4] = clip3(0, maxv, m4 - delta);
      if (filt_p) s[2] = clip3(0, maxv, m2 + clip3(-tc2, tc2, (((m1 + m3 + 1) >> 1) - m2 + delta) >> 1));
      if (filt_q) s[5] = clip3(0, maxv, m5 + clip3(-tc2, tc2, (((m6 + m4 + 1) >> 1) - m5 - delta) >> 1));
    }
  }
}

__device__ inline void unpack8(const uint4 v, int (&s)[8]) {
  s[0] = v.x & 0xffff; s[1] = v.x >> 16; s[2] = v.y & 0xffff; s[3] = v.y >> 16;
  s[4] = v.z & 0xffff; s[5] = v.z >> 16; s[6] = v.w & 0xffff; s[7] = v.w >> 16;
}
__device__ inline uint4 pack8(const int (&s)[8]) {
  return make_uint4((uint32_t)s[0] | ((uint32_t)s[1] << 16), (uint32_t)s[2] | ((uint32_t)s[3] << 16),
                    (uint32_t)s[4] | ((uint32_t)s[5] << 16), (uint32_t)s[6] | ((uint32_t)s[7] << 16));
}
struct __attribute__((aligned(8))) U4a8 { uint32_t x, y, z, w; };    // 16 bytes at 8-byte alignment

// luma decisions for one 4-line unit (xEdgeFilterLuma :587-650); l[i] = line i, 8 samples across the edge
__device__ inline void filter_luma_unit(int (&l)[4][8], int bs, int qp, int tc_offset_div2, int beta_offset_div2, int bd) {
  const int scale = 1 << (bd - 8);
  const int tc = c_tc_table[clip3(0, 53, qp + 2 * (bs - 1) + (tc_offset_div2 << 1))] * scale;
  const int beta = c_beta_table[clip3(0, 51, qp + (beta_offset_div2 << 1))] * scale;
  const int side = (beta + (beta >> 1)) >> 3;
  const int dp0 = abs(l[0][1] - 2 * l[0][2] + l[0][3]), dq0 = abs(l[0][4] - 2 * l[0][5] + l[0][6]);
  const int dp3 = abs(l[3][1] - 2 * l[3][2] + l[3][3]), dq3 = abs(l[3][4] - 2 * l[3][5] + l[3][6]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
  if (d >= beta) return;
  const bool fp = (dp0 + dp3) < side, fq = (dq0 + dq3) < side;
  const bool s0 = (abs(l[0][0] - l[0][3]) + abs(l[0][7] - l[0][4]) < (beta >> 3)) && (2 * d0 < (beta >> 2)) &&
                  (abs(l[0][3] - l[0][4]) < ((tc * 5 + 1) >> 1));
  const bool s3 = (abs(l[3][0] - l[3][3]) + abs(l[3][7] - l[3][4]) < (beta >> 3)) && (2 * d3 < (beta >> 2)) &&
                  (abs(l[3][3] - l[3][4]) < ((tc * 5 + 1) >> 1));
  const int maxv = (1 << bd) - 1;
#pragma unroll
  for (int i = 0; i < 4; i++) filter_luma_line(l[i], tc, s0 && s3, tc * 10, fp, fq, maxv);
}

// chroma tc for one component (xEdgeFilterChroma :759-775, 4:2:0)
__device__ inline int chroma_tc(int qp_avg, int pps_off, int tc_offset_div2, int bd) {
  int qp = qp_avg + pps_off;
  if (qp >= 58) qp -= 6;
  else if (qp >= 0) qp = c_chroma_scale_420_dbk[qp];
  return c_tc_table[clip3(0, 53, qp + 2 + (tc_offset_div2 << 1))] * (1 << (bd - 8));     // Bs == 2: + DEFAULT_INTRA_TC_OFFSET
}

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
  if (DIR == 0 ? gx == 0 : gy == 0) return;
  const BlkInfo q = ld_blk(P.blk + (size_t)gy * P.grid_w + gx);
  if (!(q.edge & (DIR == 0 ? BE_VER_FILTER : BE_HOR_FILTER))) return;
  const BlkInfo p = ld_blk(DIR == 0 ? P.blk + (size_t)gy * P.grid_w + gx - 1 : P.blk + (size_t)(gy - 1) * P.grid_w + gx);
  const int bs = boundary_strength(p, q, (q.edge & (DIR == 0 ? BE_VER_TRANSFORM : BE_HOR_TRANSFORM)) != 0);
  if (bs == 0) return;
  const SliceDev* slp = P.slices + q.slice;                         // offsets come from the Q side's slice (:565-566)
  const int tc_off = ldg(&slp->tc_offset_div2), beta_off = ldg(&slp->beta_offset_div2);
  const int qp = ((int)p.qp + (int)q.qp + 1) >> 1;
  const int pitch = P.pitch[0];
  int16_t* base = P.rec[0] + (size_t)y * pitch + x;
  int l[4][8];
  if (DIR == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x4 v = ldg4_a8(base + (size_t)i * pitch - 4);
      unpack8(make_uint4(v.x, v.y, v.z, v.w), l[i]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const u32x2 v = ldg2(base + (ptrdiff_t)(r - 4) * pitch);
      l[0][r] = v.x & 0xffff; l[1][r] = v.x >> 16; l[2][r] = v.y & 0xffff; l[3][r] = v.y >> 16;
    }
  }
  filter_luma_unit(l, bs, qp, tc_off, beta_off, P.bd[0]);
  if (DIR == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint4 v = pack8(l[i]);
      u32x4 o = {v.x, v.y, v.z, v.w};
      stg4_a8(base + (size_t)i * pitch - 4, o);
    }
  } else {
#pragma unroll
    for (int r = 1; r < 7; r++)
    {
      u32x2 o = {(uint32_t)l[0][r] | ((uint32_t)l[1][r] << 16), (uint32_t)l[2][r] | ((uint32_t)l[3][r] << 16)};
      stg2(base + (ptrdiff_t)(r - 4) * pitch, o);
    }
  }
  // chroma: Bs 2 only, edges on the 8-sample chroma grid = 16-sample luma grid (:225-229, :684-692, :727)
  if (bs == 2 && ((DIR == 0 ? x : y) & 15) == 0) {
    const int cp = P.pitch[1];
    const int maxc = (1 << P.bd[1]) - 1;
#pragma unroll
    for (int comp = 1; comp < 3; comp++) {
      const int tc = chroma_tc(qp, ldg(comp == 1 ? &slp->pps_cb_qp_offset : &slp->pps_cr_qp_offset), tc_off, P.bd[comp]);
      int16_t* cb = P.rec[comp] + (size_t)(y >> 1) * cp + (x >> 1);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        // xPelFilterChroma (:870-891): two lines per unit
        int16_t* s = DIR == 0 ? cb + (size_t)i * cp : cb + i;
        const ptrdiff_t o = DIR == 0 ? 1 : cp;
        const int m2 = (uint16_t)ldg(s - 2 * o), m3 = (uint16_t)ldg(s - o), m4 = (uint16_t)ldg(s), m5 = (uint16_t)ldg(s + o);
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        stg(s - o, (int16_t)clip3(0, maxc, m3 + delta));
        stg(s, (int16_t)clip3(0, maxc, m4 - delta));
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
