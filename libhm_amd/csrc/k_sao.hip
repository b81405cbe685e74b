// k_sao.hip -- sample adaptive offset (TComSampleAdaptiveOffset::SAOProcess / offsetCTU / offsetBlock,
// TComSampleAdaptiveOffset.cpp:375-734).
//
// HM copies the whole deblocked picture to a temporary and filters CTU by CTU from that snapshot.  Here the
// deblocked planes (PicDev::rec) ARE the snapshot: the kernel reads them and writes the SAO planes (PicDev::sao),
// which become the picture's final planes; no copy pass exists.  One thread per 8 horizontally adjacent samples of one
// component.  HM's per-CTU loop bounds (first/last row and column skipped when a neighbouring CTU is unavailable) are
// equivalent to one per-sample rule: a sample is modified iff both samples it is compared with lie in the current CTU
// or in a neighbouring CTU whose availability bit is set (derivation in DESIGN.md "SAO bounds").
#include "hmgpu_dev.h"

namespace hmgpu {

__device__ inline int sgn3(int v) { return (v > 0) - (v < 0); }

__global__ void __launch_bounds__(256) k_sao(const PicDev* __restrict__ pics, Batch b) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
  int row = blockIdx.y * 4 + (threadIdx.x >> 6);
  int comp = 0;
  if (row >= P.height) { row -= P.height; comp = 1; if (row >= (P.height >> 1)) { row -= P.height >> 1; comp = 2; } }
  const int cs = comp ? 1 : 0;
  const int w = P.width >> cs, h = P.height >> cs;
  if (x >= w || row >= h) return;
  const int pitch = P.pitch[comp];
  const int16_t* __restrict__ src = P.rec[comp];
  int16_t* __restrict__ dst = P.sao[comp];
  const int ctb = (1 << P.log2ctu) >> cs;                       // CTB size in this component
  const int cx = x / ctb, cy = row / ctb;
  const SaoDev* __restrict__ pp = &P.saoprm[((size_t)cy * P.ctus_w + cx) * 3 + comp];
  struct { int type; unsigned avail; } prm = {pp->type, pp->avail};   // offsets are indexed per sample: read through L1
  const uint4 cur = *reinterpret_cast<const uint4*>(src + (size_t)row * pitch + x);
  if (prm.type < 0) { *reinterpret_cast<uint4*>(dst + (size_t)row * pitch + x) = cur; return; }
  int c[8] = {(int)(cur.x & 0xffff), (int)(cur.x >> 16), (int)(cur.y & 0xffff), (int)(cur.y >> 16),
              (int)(cur.z & 0xffff), (int)(cur.z >> 16), (int)(cur.w & 0xffff), (int)(cur.w >> 16)};
  const int bd = P.bd[comp];
  const int maxv = (1 << bd) - 1;
  int o[8];
  if (prm.type == HMGPU_SAO_BO) {
    const int shift = bd - 5;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = clip3(0, maxv, c[i] + pp->offset[c[i] >> shift]);
  } else {
    // neighbour direction a = (dx, dy), b = (-dx, -dy)
    const int dx = prm.type == HMGPU_SAO_EO_90 ? 0 : (prm.type == HMGPU_SAO_EO_45 ? 1 : -1);
    const int dy = prm.type == HMGPU_SAO_EO_0 ? 0 : -1;
    // CTB bounds in component samples (clipped to the picture, as offsetCTU does: :679-682)
    const int x0 = cx * ctb, y0 = cy * ctb;
    const int x1 = min(x0 + ctb, w) - 1, y1 = min(y0 + ctb, h) - 1;
    const int ya = row + dy, yb = row - dy;
    const int16_t* ra = src + (size_t)clip3(0, h - 1, ya) * pitch;
    const int16_t* rb = src + (size_t)clip3(0, h - 1, yb) * pitch;
    // vertical class of the two neighbour rows: 0 above the CTB, 1 inside, 2 below
    const int va = ya < y0 ? 0 : (ya > y1 ? 2 : 1), vb = yb < y0 ? 0 : (yb > y1 ? 2 : 1);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int xa = x + i + dx, xb = x + i - dx;
      const int ha = xa < x0 ? 0 : (xa > x1 ? 2 : 1), hb = xb < x0 ? 0 : (xb > x1 ? 2 : 1);
      // availability bit of the CTU a position falls into: order L,R,A,B,AL,AR,BL,BR (SaoDev::avail); inside = always
      // index by (v,h): (0,0) AL=4 (0,1) A=2 (0,2) AR=5 (1,0) L=0 (1,1) inside (1,2) R=1 (2,0) BL=6 (2,1) B=3 (2,2) BR=7
      const int bit_a = va == 1 ? (ha == 0 ? 0 : (ha == 2 ? 1 : 8)) : (va == 0 ? (ha == 0 ? 4 : (ha == 2 ? 5 : 2)) : (ha == 0 ? 6 : (ha == 2 ? 7 : 3)));
      const int bit_b = vb == 1 ? (hb == 0 ? 0 : (hb == 2 ? 1 : 8)) : (vb == 0 ? (hb == 0 ? 4 : (hb == 2 ? 5 : 2)) : (hb == 0 ? 6 : (hb == 2 ? 7 : 3)));
      const unsigned av = (unsigned)prm.avail | 0x100u;
      const bool ok = ((av >> bit_a) & 1) && ((av >> bit_b) & 1);
      const int sa = (uint16_t)ra[clip3(0, w - 1, xa)], sb = (uint16_t)rb[clip3(0, w - 1, xb)];
      const int et = sgn3(c[i] - sa) + sgn3(c[i] - sb);
      o[i] = ok ? clip3(0, maxv, c[i] + pp->offset[2 + et]) : c[i];
    }
  }
  *reinterpret_cast<uint4*>(dst + (size_t)row * pitch + x) =
      make_uint4((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16),
                 (uint32_t)o[4] | ((uint32_t)o[5] << 16), (uint32_t)o[6] | ((uint32_t)o[7] << 16));
}

void launch_sao(const PicDev* pics, const Batch& b, int width, int height, hipStream_t s) {
  dim3 grid((unsigned)((width / 8 + 63) / 64), (unsigned)((2 * height + 3) / 4), (unsigned)b.n);
  hipLaunchKernelGGL(k_sao, grid, dim3(256), 0, s, pics, b);
}

}  // namespace hmgpu
