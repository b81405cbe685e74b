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
#include "filter_core.h"

namespace hmgpu {

// edge offset of one row of 8 samples, compile-time direction (DX, DY in {-1,0,1}); everything stays packed (2 samples / register)
// src: the component's sample (0, 0); comp > 0: its samples lie kCStep elements apart (ldc8)
template <int DX, int DY>
__device__ inline void sao_eo_row(const int16_t* __restrict__ src, int comp, int pitch, int w, int h, int x, int row, const u32x4 cur, uint32_t off_lo,
                                  uint32_t off_hi, unsigned av, int x0, int y0, int x1, int y1, int maxv, uint32_t (&out)[4]) {
  const int ya = row + DY, yb = row - DY;
  const int16_t* ra = src + (size_t)clip3(0, h - 1, ya) * pitch;
  const int16_t* rb = src + (size_t)clip3(0, h - 1, yb) * pitch;
  const int st = comp ? kCStep : 1;
  uint32_t na[4], nb[4];
  {
    const u32x4 ea = DY == 0 ? cur : (comp ? ldc8(ra, x, comp) : ldg4(ra + x)), eb = DY == 0 ? cur : (comp ? ldc8(rb, x, comp) : ldg4(rb + x));
    uint32_t la = 0, raa = 0, lb = 0, rbb = 0;
    if constexpr (DX < 0) { la = (uint16_t)ldg(ra + st * max(x - 1, 0)); rbb = (uint16_t)ldg(rb + st * min(x + 8, w - 1)); }
    if constexpr (DX > 0) { raa = (uint16_t)ldg(ra + st * min(x + 8, w - 1)); lb = (uint16_t)ldg(rb + st * max(x - 1, 0)); }
    shifted<DX>(ea, la, raa, na);
    shifted<-DX>(eb, lb, rbb, nb);
  }
  sao_eo_core<DX, DY>(x, row, cur, na, nb, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out);
}

// One wave = one 64x8 luma block (8 lanes across, 8 rows) or one 32x16 chroma block (4 lanes across, 16 rows): always
// inside ONE CTB, so the SAO type is wave-uniform and the type switch below costs no divergence.
__global__ void __launch_bounds__(256) k_sao(const PicDev* __restrict__ pics, Batch b, int luma_waves, int chroma_waves) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int comp = 0;
  if (wid >= luma_waves) { wid -= luma_waves; comp = 1; if (wid >= chroma_waves) { wid -= chroma_waves; comp = 2; if (wid >= chroma_waves) return; } }
  const int sx = comp ? P.csx : 0, sy = comp ? P.csy : 0;    // subsampling of the component (4:2:0: 1, 1; 4:2:2: 1, 0; 4:4:4: 0, 0)
  const int w = P.width >> sx, h = P.height >> sy;
  const int bw = 64 >> sx, bh = 8 << sx;                    // block covered by the wave (samples): 64 x 8, or 32 x 16 across a subsampled direction
  const int blocks_x = (w + bw - 1) / bw;
  const int bx = wid % blocks_x, by = wid / blocks_x;
  const int lanes_x = bw / 8;
  const int x = bx * bw + (lane % lanes_x) * 8;
  const int row = by * bh + lane / lanes_x;
  if (x >= w || row >= h) return;
  const int pitch = P.pitch[comp];
  const int16_t* __restrict__ src = P.rec[comp];
  int16_t* __restrict__ dst = P.sao[comp];
  const int log2ctb_x = P.log2ctu - sx, log2ctb_y = P.log2ctu - sy;
  const int cx = x >> log2ctb_x, cy = row >> log2ctb_y;
  const uint32_t* pw = reinterpret_cast<const uint32_t*>(P.saoprm + ((size_t)cy * P.ctus_w + cx) * 3 + comp);
  const uint32_t w0 = ldg(pw);
  const int type = (int)(int8_t)(w0 & 0xff);
  const u32x4 cur = comp ? ldc8(src + (size_t)row * pitch, x, comp) : ldg4(src + (size_t)row * pitch + x);
  if (type < 0) { if (comp) stc8(dst + (size_t)row * pitch, x, cur); else stg4(dst + (size_t)row * pitch + x, cur); return; }
  const uint32_t off_lo = ldg(pw + 1), off_hi = ldg(pw + 2);            // off[0..3], off[4..7]
  const int bd = P.bd[comp];
  const int maxv = (1 << bd) - 1;
  uint32_t out[4];
  if (type == HMGPU_SAO_BO) {
    // band k = (sample >> (bd-5)) - first band (mod 32); bands 0..3 carry offsets, every other band maps to table entry 4 (= 0)
    const int shift = bd - 5, band0 = (w0 >> 8) & 0xff;
    const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t lo = c[j] & 0xffffu, hi = c[j] >> 16;
      const uint32_t k0 = min(((lo >> shift) - band0) & 31u, 4u), k1 = min(((hi >> shift) - band0) & 31u, 4u);
      const s16x2 off = lut_offsets(k0 | (k1 << 16), off_lo, 0u);
      out[j] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(as_s16x2(c[j]) + off, splat(0)), splat(maxv)));
    }
  } else {
    const int x0 = cx << log2ctb_x, y0 = cy << log2ctb_y;
    const int x1 = min(x0 + (1 << log2ctb_x), w) - 1, y1 = min(y0 + (1 << log2ctb_y), h) - 1;     // CTB clipped to the picture (offsetCTU :679-682)
    const unsigned av = w0 >> 16;
    switch (type) {                                                      // a = (x+DX, y+DY), b = (x-DX, y-DY)
      case HMGPU_SAO_EO_0:   sao_eo_row<-1, 0>(src, comp, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
      case HMGPU_SAO_EO_90:  sao_eo_row<0, -1>(src, comp, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
      case HMGPU_SAO_EO_135: sao_eo_row<-1, -1>(src, comp, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
      default:               sao_eo_row<1, -1>(src, comp, pitch, w, h, x, row, cur, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out); break;
    }
  }
  if (P.any_nofilt) {
    uint32_t m[4];
    sao_exempt_mask(P, comp, x, row, m);
    const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = (c[j] & m[j]) | (out[j] & ~m[j]);
  }
  u32x4 res = {out[0], out[1], out[2], out[3]};
  if (comp) stc8(dst + (size_t)row * pitch, x, res); else stg4(dst + (size_t)row * pitch + x, res);
}

// TComPicYuv::extendPicBorder (TComPicYuv.cpp:173-217): replicate the edge samples of the picture's FINAL planes into the
// margins so that motion compensation of later pictures never has to clamp coordinates.  One thread per 16-byte piece of
// margin (8 samples): side bands first (every row of the padded plane, left and right), then the bands above and below the
// picture columns.  Every thread reads only samples of the visible picture, so there is no ordering between margin writes.
// E = int16 elements per picture element: 1 luma (an element = a sample), kCStep chroma (an element = the (Cb, Cr) pair of a position: the two
// components share a plane, hmgpu_dev.h "chroma planes")
template <int E>
__device__ inline void extend_plane(int16_t* pl, int w, int h, int mx, int my, int pitch, int idx) {
  constexpr int VE = 8 / E;                                  // elements per 16-byte vector
  auto edge = [&](int y, int x) -> uint32_t {                // element (x, y) as a dword pattern
    if constexpr (E == 1) return (uint16_t)ldg(pl + (ptrdiff_t)y * pitch + x) * 0x10001u;
    else return ldg(reinterpret_cast<const uint32_t*>(pl + (ptrdiff_t)y * pitch + E * x));
  };
  const int side_vecs_per_row = 2 * mx / VE;                 // left + right band
  const int rows = h + 2 * my;
  if (idx < side_vecs_per_row * rows) {
    const int y = idx / side_vecs_per_row - my, k = idx % side_vecs_per_row;
    const bool right = k >= mx / VE;
    const int x = right ? w + (k - mx / VE) * VE : -mx + k * VE;
    const uint32_t v = edge(clip3(0, h - 1, y), right ? w - 1 : 0);
    u32x4 o = {v, v, v, v};
    stg4(pl + (ptrdiff_t)y * pitch + E * x, o);              // (widths are multiples of 8 luma samples: both bands start on 16-byte boundaries)
  } else {
    idx -= side_vecs_per_row * rows;
    const int vecs_per_row = (w + VE - 1) / VE;
    const int r = idx / vecs_per_row;                         // 0 .. 2*my-1
    if (r >= 2 * my) return;
    const int y = r < my ? r - my : h + (r - my);
    const int x = (idx % vecs_per_row) * VE;
    const int sy = r < my ? 0 : h - 1;
    // (a picture whose width is not a multiple of a vector lets the last vector run into the right band: same values)
    u32x4 o = ldg4(pl + (ptrdiff_t)sy * pitch + E * x);
    if (x + VE > w) {
      const uint32_t e = edge(sy, w - 1);
      uint32_t t[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if constexpr (E == 1) {
          if (x + 2 * j >= w) t[j] = e;
          else if (x + 2 * j + 1 >= w) t[j] = (t[j] & 0xffffu) | (e & 0xffff0000u);
        } else if (x + j >= w) t[j] = e;
      }
      o = (u32x4){t[0], t[1], t[2], t[3]};
    }
    stg4(pl + (ptrdiff_t)y * pitch + E * x, o);
  }
}
__global__ void __launch_bounds__(256) k_extend(const PicDev* __restrict__ pics, Batch b, int luma_vecs, int chroma_vecs) {
  const PicDev& P = pics[b.pic[blockIdx.z]];
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < luma_vecs) { extend_plane<1>(P.sao_applied ? P.sao[0] : P.rec[0], P.width, P.height, P.mx[0], P.my[0], P.pitch[0], idx); return; }
  idx -= luma_vecs;
  if (idx < chroma_vecs) extend_plane<kCStep>(P.sao_applied ? P.sao[1] : P.rec[1], P.width >> P.csx, P.height >> P.csy, P.mx[1], P.my[1], P.pitch[1], idx);
}

void launch_extend(const PicDev* pics, const Batch& b, int width, int height, int mx, int my, int csx, int csy, hipStream_t s) {
  // luma margins (mx, my); chroma margins follow the subsampling (hmgpu_create), a chroma element is a (Cb, Cr) pair = 4 bytes
  const int luma = (2 * mx / 8) * (height + 2 * my) + ((width + 7) / 8) * 2 * my;
  const int cw = width >> csx, chh = height >> csy, cmx = mx >> csx, cmy = my >> csy, ve = 8 / kCStep;
  const int chroma = (2 * cmx / ve) * (chh + 2 * cmy) + ((cw + ve - 1) / ve) * 2 * cmy;
  dim3 grid((unsigned)((luma + chroma + 255) / 256), 1, (unsigned)b.n);
  hipLaunchKernelGGL(k_extend, grid, dim3(256), 0, s, pics, b, luma, chroma);
}

void launch_sao(const PicDev* pics, const Batch& b, int width, int height, int csx, int csy, hipStream_t s) {
  const int luma = ((width + 63) / 64) * ((height + 7) / 8);
  const int cbw = 64 >> csx, cbh = 8 << csx;                 // (k_sao: the block of a chroma wave)
  const int chroma = (((width >> csx) + cbw - 1) / cbw) * (((height >> csy) + cbh - 1) / cbh);
  dim3 grid((unsigned)((luma + 2 * chroma + 3) / 4), 1, (unsigned)b.n);
  hipLaunchKernelGGL(k_sao, grid, dim3(256), 0, s, pics, b, luma, chroma);
}

}  // namespace hmgpu
