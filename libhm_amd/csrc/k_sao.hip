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

__device__ inline void unpack8u(const u32x4 v, int (&s)[8]) {
  s[0] = v.x & 0xffff; s[1] = v.x >> 16; s[2] = v.y & 0xffff; s[3] = v.y >> 16;
  s[4] = v.z & 0xffff; s[5] = v.z >> 16; s[6] = v.w & 0xffff; s[7] = v.w >> 16;
}

// the 8 samples at x+dx .. x+7+dx of row `r` (dx in {-1,0,1}); columns are clamped into the picture (the clamped
// values are only ever used for samples that the availability rule leaves untouched)
__device__ inline void row_shifted(const int16_t* __restrict__ r, int x, int w, int dx, int (&s)[8]) {
  int v[8];
  unpack8u(ldg4(r + x), v);
  const int l = (uint16_t)ldg(r + max(x - 1, 0)), rr = (uint16_t)ldg(r + min(x + 8, w - 1));
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
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(P.saoprm + ((size_t)cy * P.ctus_w + cx) * 3 + comp);
  const uint32_t w0 = ldg(pw);
  const int type = (int)(int8_t)(w0 & 0xff);
  const u32x4 cur = ldg4(src + (size_t)row * pitch + x);
  if (type < 0) { stg4(dst + (size_t)row * pitch + x, cur); return; }
  const uint32_t off_lo = ldg(pw + 1), off_hi = ldg(pw + 2);            // off[0..3], off[4..7]
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
  u32x4 res = {(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16),
               (uint32_t)o[4] | ((uint32_t)o[5] << 16), (uint32_t)o[6] | ((uint32_t)o[7] << 16)};
  stg4(dst + (size_t)row * pitch + x, res);
}

// TComPicYuv::extendPicBorder (TComPicYuv.cpp:173-217): replicate the edge samples of the picture's FINAL planes into the
// margins so that motion compensation of later pictures never has to clamp coordinates.  One thread per margin sample
// pair; every thread reads only samples of the visible picture, so there is no ordering between margin writes.
__global__ void __launch_bounds__(256) k_extend(const PicDev* __restrict__ pics, Batch b, int total_luma_pairs, int total_chroma_pairs) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  int idx = blockIdx.x * 256 + threadIdx.x;
  int comp = 0;
  if (idx >= total_luma_pairs) { idx -= total_luma_pairs; comp = 1; if (idx >= total_chroma_pairs) { idx -= total_chroma_pairs; comp = 2; if (idx >= total_chroma_pairs) return; } }
  const int cs = comp ? 1 : 0;
  const int w = P.width >> cs, h = P.height >> cs, mx = P.mx[comp], my = P.my[comp], pitch = P.pitch[comp];
  int16_t* pl = P.sao_applied ? P.sao[comp] : P.rec[comp];
  // margin region in pairs of samples: first the left+right bands of the rows [-my, h+my), then the top+bottom bands over [0, w)
  const int side_pairs_per_row = mx;                         // (mx left + mx right) / 2
  const int rows = h + 2 * my;
  int x, y;
  if (idx < side_pairs_per_row * rows) {
    y = idx / side_pairs_per_row - my;
    const int k = (idx % side_pairs_per_row) * 2;            // 0 .. 2*mx-2
    x = k < mx ? k - mx : w + (k - mx);
  } else {
    idx -= side_pairs_per_row * rows;
    const int pairs_per_row = w >> 1;
    const int r = idx / pairs_per_row;                       // 0 .. 2*my-1
    if (r >= 2 * my) return;
    y = r < my ? r - my : h + (r - my);
    x = (idx % pairs_per_row) * 2;
  }
  const int sy = clip3(0, h - 1, y);
  const uint32_t a = (uint16_t)ldg(pl + (ptrdiff_t)sy * pitch + clip3(0, w - 1, x));
  const uint32_t c = (uint16_t)ldg(pl + (ptrdiff_t)sy * pitch + clip3(0, w - 1, x + 1));
  stg(reinterpret_cast<uint32_t*>(pl + (ptrdiff_t)y * pitch + x), a | (c << 16));
}

void launch_extend(const PicDev* pics, const Batch& b, int width, int height, int mx, int my, hipStream_t s) {
  // luma margins (mx, my); chroma margins are half of them
  const int luma = mx * (height + 2 * my) + (width / 2) * 2 * my;
  const int cw = width / 2, chh = height / 2, cmx = mx / 2, cmy = my / 2;
  const int chroma = cmx * (chh + 2 * cmy) + (cw / 2) * 2 * cmy;
  dim3 grid((unsigned)((luma + 2 * chroma + 255) / 256), 1, (unsigned)b.n);
  hipLaunchKernelGGL(k_extend, grid, dim3(256), 0, s, pics, b, luma, chroma);
}

void launch_sao(const PicDev* pics, const Batch& b, int width, int height, hipStream_t s) {
  dim3 grid((unsigned)((width / 8 + 63) / 64), (unsigned)((2 * height + 3) / 4), (unsigned)b.n);
  hipLaunchKernelGGL(k_sao, grid, dim3(256), 0, s, pics, b);
}

}  // namespace hmgpu
