// k_sao.hip -- sample adaptive offset (TComSampleAdaptiveOffset::SAOProcess / offsetCTU / offsetBlock,
// TComSampleAdaptiveOffset.cpp:375-734).
//
// HM copies the whole deblocked picture to a temporary and filters CTU by CTU from that snapshot.  Here the
// deblocked planes (PicDev::rec) ARE the snapshot: the kernel reads them and writes the SAO planes (PicDev::sao),
// which become the picture's final planes; no copy pass exists.  One thread per 8 horizontally adjacent samples of one
// component: 16-byte loads of the current row and of the two neighbour rows, one 16-byte store.  HM's per-CTU loop
// bounds (first/last row and column skipped when a neighbouring CTU is unavailable) are equivalent to one per-sample
// rule: a sample is modified iff both samples it is compared with lie in the current CTU or in a neighbouring CTU
// whose availability bit is set (derivation in DESIGN.md "SAO bounds"); only the first and last sample of a thread's
// group can face another CTU horizontally, so the rule costs a handful of scalar-like operations per thread.
#include "hmgpu_dev.h"

namespace hmgpu {

__device__ inline void unpack8u(const uint4 v, int (&s)[8]) {
  s[0] = v.x & 0xffff; s[1] = v.x >> 16; s[2] = v.y & 0xffff; s[3] = v.y >> 16;
  s[4] = v.z & 0xffff; s[5] = v.z >> 16; s[6] = v.w & 0xffff; s[7] = v.w >> 16;
}

// the 8 samples at x+dx .. x+7+dx of row `r` (dx in {-1,0,1}); columns are clamped into the picture (the clamped
// values are only ever used for samples that the availability rule leaves untouched)
__device__ inline void row_shifted(const int16_t* __restrict__ r, int x, int w, int dx, int (&s)[8]) {
  int v[8];
  unpack8u(*reinterpret_cast<const uint4*>(r + x), v);
  const int l = (uint16_t)r[max(x - 1, 0)], rr = (uint16_t)r[min(x + 8, w - 1)];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int left = i == 0 ? l : v[i - 1], right = i == 7 ? rr : v[i + 1];
    s[i] = dx == 0 ? v[i] : (dx < 0 ? left : right);
  }
}

// availability bit (SaoDev::avail order L,R,A,B,AL,AR,BL,BR; 8 = inside the CTB) of the CTU that holds a position with
// vertical class v (0 above, 1 inside, 2 below) and horizontal class hcls (0 left, 1 inside, 2 right)
__device__ inline int region_bit(int v, int hcls) {
  return v == 1 ? (hcls == 0 ? 0 : (hcls == 2 ? 1 : 8)) : (v == 0 ? (hcls == 0 ? 4 : (hcls == 2 ? 5 : 2)) : (hcls == 0 ? 6 : (hcls == 2 ? 7 : 3)));
}

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
  const int log2ctb = P.log2ctu - cs;
  const int cx = x >> log2ctb, cy = row >> log2ctb;
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(&P.saoprm[((size_t)cy * P.ctus_w + cx) * 3 + comp]);
  const uint32_t w0 = pw[0];
  const int type = (int)(int8_t)(w0 & 0xff);
  const uint4 cur = *reinterpret_cast<const uint4*>(src + (size_t)row * pitch + x);
  if (type < 0) { *reinterpret_cast<uint4*>(dst + (size_t)row * pitch + x) = cur; return; }
  const uint32_t off_lo = pw[1], off_hi = pw[2];            // off[0..3], off[4..7]
  int c[8], o[8];
  unpack8u(cur, c);
  const int bd = P.bd[comp];
  const int maxv = (1 << bd) - 1;
  if (type == HMGPU_SAO_BO) {
    const int shift = bd - 5, band0 = (w0 >> 16) & 0xff;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned k = ((c[i] >> shift) - band0) & 31;
      const int off = k < 4 ? (int)(int8_t)(off_lo >> (8 * k)) : 0;
      o[i] = clip3(0, maxv, c[i] + off);
    }
  } else {
    // neighbour a = (dx, dy), neighbour b = (-dx, -dy)
    const int dx = type == HMGPU_SAO_EO_90 ? 0 : (type == HMGPU_SAO_EO_45 ? 1 : -1);
    const int dy = type == HMGPU_SAO_EO_0 ? 0 : -1;
    const int ctb = 1 << log2ctb;
    const int x0 = cx << log2ctb, y0 = cy << log2ctb;
    const int x1 = min(x0 + ctb, w) - 1, y1 = min(y0 + ctb, h) - 1;        // CTB clipped to the picture (offsetCTU :679-682)
    const int ya = row + dy, yb = row - dy;
    int sa[8], sb[8];
    row_shifted(src + (size_t)clip3(0, h - 1, ya) * pitch, x, w, dx, sa);
    row_shifted(src + (size_t)clip3(0, h - 1, yb) * pitch, x, w, -dx, sb);
    const int va = ya < y0 ? 0 : (ya > y1 ? 2 : 1), vb = yb < y0 ? 0 : (yb > y1 ? 2 : 1);
    const unsigned av = ((w0 >> 8) & 0xff) | 0x100u;
    // interior samples of the group compare with positions in the CTB's own columns
    const bool mid_ok = ((av >> region_bit(va, 1)) & 1) && ((av >> region_bit(vb, 1)) & 1);
    // sample 0 may face the left CTU column, the last sample the right one
    const int ha0 = (x + dx) < x0 ? 0 : 1, hb0 = (x - dx) < x0 ? 0 : 1;
    const int ha7 = (x + 7 + dx) > x1 ? 2 : 1, hb7 = (x + 7 - dx) > x1 ? 2 : 1;
    const bool ok0 = ((av >> region_bit(va, ha0)) & 1) && ((av >> region_bit(vb, hb0)) & 1);
    const bool ok7 = ((av >> region_bit(va, ha7)) & 1) && ((av >> region_bit(vb, hb7)) & 1);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int et = ((c[i] > sa[i]) - (c[i] < sa[i])) + ((c[i] > sb[i]) - (c[i] < sb[i])) + 2;       // 0..4
      const int off = et < 4 ? (int)(int8_t)(off_lo >> (8 * et)) : (int)(int8_t)(off_hi & 0xff);
      const bool ok = i == 0 ? ok0 : (i == 7 ? ok7 : mid_ok);
      o[i] = ok ? clip3(0, maxv, c[i] + off) : c[i];
    }
    // a picture narrower than the group (chroma width not a multiple of 8): the "last" sample is x1, not x+7
    if (x + 7 > x1) {
#pragma unroll
      for (int i = 1; i < 7; i++) {
        if (x + i == x1) {
          const int ha = (x + i + dx) > x1 ? 2 : 1, hb = (x + i - dx) > x1 ? 2 : 1;
          const bool ok = ((av >> region_bit(va, ha)) & 1) && ((av >> region_bit(vb, hb)) & 1);
          const int et = ((c[i] > sa[i]) - (c[i] < sa[i])) + ((c[i] > sb[i]) - (c[i] < sb[i])) + 2;
          const int off = et < 4 ? (int)(int8_t)(off_lo >> (8 * et)) : (int)(int8_t)(off_hi & 0xff);
          o[i] = ok ? clip3(0, maxv, c[i] + off) : c[i];
        }
      }
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
