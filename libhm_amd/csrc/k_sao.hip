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

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ inline s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ inline uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline s16x2 splat(int v) { return (s16x2){(short)v, (short)v}; }

// availability bit (SaoDev::avail order L,R,A,B,AL,AR,BL,BR; 8 = inside the CTB) of the CTU that holds a position with
// vertical class v (0 above, 1 inside, 2 below) and horizontal class hcls (0 left, 1 inside, 2 right)
__device__ inline int region_bit(int v, int hcls) {
  return v == 1 ? (hcls == 0 ? 0 : (hcls == 2 ? 1 : 8)) : (v == 0 ? (hcls == 0 ? 4 : (hcls == 2 ? 5 : 2)) : (hcls == 0 ? 6 : (hcls == 2 ? 7 : 3)));
}

// neighbour samples x+DX .. x+7+DX of a row as four packed pairs; `e` = the row's 8 samples, l / r = samples x-1 / x+8
template <int DX>
__device__ inline void shifted(const u32x4 e, uint32_t l, uint32_t r, uint32_t (&n)[4]) {
  if constexpr (DX == 0) { n[0] = e.x; n[1] = e.y; n[2] = e.z; n[3] = e.w; }
  else if constexpr (DX < 0) {
    n[0] = (e.x << 16) | l; n[1] = __builtin_amdgcn_alignbit(e.y, e.x, 16);
    n[2] = __builtin_amdgcn_alignbit(e.z, e.y, 16); n[3] = __builtin_amdgcn_alignbit(e.w, e.z, 16);
  } else {
    n[0] = __builtin_amdgcn_alignbit(e.y, e.x, 16); n[1] = __builtin_amdgcn_alignbit(e.z, e.y, 16);
    n[2] = __builtin_amdgcn_alignbit(e.w, e.z, 16); n[3] = (e.w >> 16) | (r << 16);
  }
}

// offsets by table index (two indices 0..7 packed as 16-bit halves) -> two sign-extended 16-bit offsets.  v_perm_b32 does
// the 8-entry byte-table lookup for both halves at once: indices are moved to the odd bytes so that the second v_perm can
// replicate the sign bits (selector codes 8 / 9 = sign of byte 1 / 3).
__device__ inline s16x2 lut_offsets(uint32_t idx_pk, uint32_t tab_lo, uint32_t tab_hi) {
  const uint32_t looked = __builtin_amdgcn_perm(tab_hi, tab_lo, idx_pk << 8);   // bytes 1,3 = table[idx]; bytes 0,2 = table[0] (unused)
  return as_s16x2(__builtin_amdgcn_perm(0u, looked, 0x09030801u));
}

// edge offset of one row of 8 samples, compile-time direction (DX, DY in {-1,0,1}); everything stays packed (2 samples / register)
template <int DX, int DY>
__device__ inline void sao_eo_row(const int16_t* __restrict__ src, int pitch, int w, int h, int x, int row, const u32x4 cur, uint32_t off_lo,
                                  uint32_t off_hi, unsigned av, int x0, int y0, int x1, int y1, int maxv, uint32_t (&out)[4]) {
  const int ya = row + DY, yb = row - DY;
  const int16_t* ra = src + (size_t)clip3(0, h - 1, ya) * pitch;
  const int16_t* rb = src + (size_t)clip3(0, h - 1, yb) * pitch;
  uint32_t na[4], nb[4];
  {
    const u32x4 ea = DY == 0 ? cur : ldg4(ra + x), eb = DY == 0 ? cur : ldg4(rb + x);
    uint32_t la = 0, raa = 0, lb = 0, rbb = 0;
    if constexpr (DX < 0) { la = (uint16_t)ldg(ra + max(x - 1, 0)); rbb = (uint16_t)ldg(rb + min(x + 8, w - 1)); }
    if constexpr (DX > 0) { raa = (uint16_t)ldg(ra + min(x + 8, w - 1)); lb = (uint16_t)ldg(rb + max(x - 1, 0)); }
    shifted<DX>(ea, la, raa, na);
    shifted<-DX>(eb, lb, rbb, nb);
  }
  // availability: interior samples face positions in the CTB's own columns; sample 0 / the last sample may face the
  // left / right CTU column
  const int va = ya < y0 ? 0 : (ya > y1 ? 2 : 1), vb = yb < y0 ? 0 : (yb > y1 ? 2 : 1);
  const bool mid_ok = ((av >> region_bit(va, 1)) & 1) && ((av >> region_bit(vb, 1)) & 1);
  const int last = min(7, x1 - x);                          // last sample of the group that lies inside the CTB / picture
  const int ha0 = (x + DX) < x0 ? 0 : 1, hb0 = (x - DX) < x0 ? 0 : 1;
  const int hal = (x + last + DX) > x1 ? 2 : 1, hbl = (x + last - DX) > x1 ? 2 : 1;
  const bool ok0 = ((av >> region_bit(va, ha0)) & 1) && ((av >> region_bit(vb, hb0)) & 1);
  const bool okl = ((av >> region_bit(va, hal)) & 1) && ((av >> region_bit(vb, hbl)) & 1);
  const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const s16x2 cc = as_s16x2(c[j]);
    const s16x2 one = splat(1), mone = splat(-1);
    const s16x2 sa = __builtin_elementwise_max(__builtin_elementwise_min(cc - as_s16x2(na[j]), one), mone);
    const s16x2 sb = __builtin_elementwise_max(__builtin_elementwise_min(cc - as_s16x2(nb[j]), one), mone);
    const uint32_t et = as_u32(sa + sb + splat(2));         // edge class 0..4 in each half
    const s16x2 off = lut_offsets(et, off_lo, off_hi);
    const s16x2 res = __builtin_elementwise_min(__builtin_elementwise_max(cc + off, splat(0)), splat(maxv));
    // per-half enable mask
    const bool ok_lo = (2 * j == 0) ? ok0 : ((2 * j == last) ? okl : mid_ok);
    const bool ok_hi = (2 * j + 1 == last) ? okl : mid_ok;
    const uint32_t m = (ok_lo ? 0xffffu : 0u) | (ok_hi ? 0xffff0000u : 0u);
    out[j] = (as_u32(res) & m) | (c[j] & ~m);
  }
}

// One wave = one 64x8 luma block (8 lanes across, 8 rows) or one 32x16 chroma block (4 lanes across, 16 rows): always
// inside ONE CTB, so the SAO type is wave-uniform and the type switch below costs no divergence.
__global__ void __launch_bounds__(256) k_sao(const PicDev* __restrict__ pics, Batch b, int luma_waves, int chroma_waves) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int comp = 0;
  if (wid >= luma_waves) { wid -= luma_waves; comp = 1; if (wid >= chroma_waves) { wid -= chroma_waves; comp = 2; if (wid >= chroma_waves) return; } }
  const int cs = comp ? 1 : 0;
  const int w = P.width >> cs, h = P.height >> cs;
  const int bw = comp ? 32 : 64, bh = comp ? 16 : 8;        // block covered by the wave (samples)
  const int blocks_x = (w + bw - 1) / bw;
  const int bx = wid % blocks_x, by = wid / blocks_x;
  const int lanes_x = bw / 8;
  const int x = bx * bw + (lane % lanes_x) * 8;
  const int row = by * bh + lane / lanes_x;
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
  const int bd = P.bd[comp];
  const int maxv = (1 << bd) - 1;
  uint32_t out[4];
  if (type == HMGPU_SAO_BO) {
    // band k = (sample >> (bd-5)) - first band (mod 32); bands 0..3 carry offsets, every other band maps to table entry 4 (= 0)
    const int shift = bd - 5, band0 = (w0 >> 16) & 0xff;
    const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t lo = c[j] & 0xffffu, hi = c[j] >> 16;
      const uint32_t k0 = min(((lo >> shift) - band0) & 31u, 4u), k1 = min(((hi >> shift) - band0) & 31u, 4u);
      const s16x2 off = lut_offsets(k0 | (k1 << 16), off_lo, 0u);
      out[j] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(as_s16x2(c[j]) + off, splat(0)), splat(maxv)));
    }
  } else {
    const int ctb = 1 << log2ctb;
    const int x0 = cx << log2ctb, y0 = cy << log2ctb;
    const int x1 = min(x0 + ctb, w) - 1, y1 = min(y0 + ctb, h) - 1;     // CTB clipped to the picture (offsetCTU :679-682)
    const unsigned av = ((w0 >> 8) & 0xff) | 0x100u;
    switch (type) {                                                      // a = (x+DX, y+DY), b = (x-DX, y-DY)
      case HMGPU_SAO_EO_0:   sao_eo_row<-1, 0>(src, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
      case HMGPU_SAO_EO_90:  sao_eo_row<0, -1>(src, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
      case HMGPU_SAO_EO_135: sao_eo_row<-1, -1>(src, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
      default:               sao_eo_row<1, -1>(src, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
    }
  }
  u32x4 res = {out[0], out[1], out[2], out[3]};
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
  const int luma = ((width + 63) / 64) * ((height + 7) / 8);
  const int chroma = ((width / 2 + 31) / 32) * ((height / 2 + 15) / 16);
  dim3 grid((unsigned)((luma + 2 * chroma + 3) / 4), 1, (unsigned)b.n);
  hipLaunchKernelGGL(k_sao, grid, dim3(256), 0, s, pics, b, luma, chroma);
}

}  // namespace hmgpu
