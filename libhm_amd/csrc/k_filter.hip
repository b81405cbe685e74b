// k_filter.hip -- deblocking of both edge directions and SAO in ONE pass over the picture
// (TComLoopFilter::loopFilterPic + TComSampleAdaptiveOffset::SAOProcess, TDecGop.cpp:165-173).
//
// The stand-alone kernels (k_dbk.hip x 2, k_sao.hip) each stream the whole picture through HBM: ~1.0 GB of real traffic per
// batch of eight 2160p pictures, at 85-100 % of the achievable bandwidth -- the remaining lever is to move fewer bytes.
// Here a workgroup owns a 64x64 luma tile (+ the 32x32 tiles of Cb and Cr), loads it ONCE with the halo the filters reach
// through, runs vertical-edge deblocking, horizontal-edge deblocking and SAO on the copy in LDS, and writes only the SAO
// planes: read ~1.5 x picture + BlkInfo, write 1 x picture.  The deblocked picture itself is never stored (with SAO on, the
// SAO planes are the picture's final planes; pictures without SAO take the stand-alone path).
//
// Halo arithmetic (luma; chroma is the same at half scale with a 1-sample filter): SAO of the tile needs deblocked samples
// one sample around it; a deblocked sample belongs to exactly one edge per direction (edges lie on the 8x8 grid, an edge
// rewrites 3 samples on each side and reads 4); so the horizontal edges y0, y0+8 .. y0+64 must be filtered over the columns
// [x0-4, x0+68), on input that has seen the vertical edges x0 .. x0+64 over the rows [y0-4, y0+68), which read the
// unfiltered columns [x0-4, x0+68).  Edge units on a tile border are computed by both neighbours, from the same unfiltered
// samples, hence identically.  Every decision and every filter is the code of the stand-alone kernels (filter_core.h).
#include "hmgpu_dev.h"
#include "filter_core.h"
#include <type_traits>

namespace hmgpu {

namespace {
constexpr int kTW = 64, kTH = 64;                 // luma tile
constexpr int kYC = kTW + 16, kYH = kTH + 8;      // luma copy: columns [x0-8, x0+72) (16-byte aligned rows), rows [y0-4, y0+68)
constexpr int kCC = kTW / 2 + 16, kCH = kTH / 2 + 4;   // chroma copy: columns [cx0-8, cx0+40), rows [cy0-2, cy0+34)
constexpr int kYW = kYC + 8, kCW = kCC + 8;       // row pitch in LDS: an odd number of 16-byte units (conflict-free 16-byte row accesses)

constexpr int kUnits = 9 * 18;                     // edge units of one direction per tile (see k_filter_fused)
struct FilterLds {
  __attribute__((aligned(16))) int16_t y[kYH][kYW];
  __attribute__((aligned(16))) int16_t c[2][kCH][kCW];
  // the edge units that really filter (Bs > 0), compacted per classifying wave: [direction][wave][slot] and their counts
  uint32_t unit[2][3][64];
  int32_t units[2][4];
};

// SAO of one group of 8 samples at (x, row) of component comp, reading the deblocked copy; (ox, oy) = picture coordinates
// of copy element [0][0]
struct SaoPrm { uint32_t w0, off_lo, off_hi; };               // SaoDev as three dwords
__device__ inline SaoPrm sao_fetch(const PicDev& P, int comp, int x, int row) {
  const int log2ctb = P.log2ctu - (comp ? 1 : 0);
  const int w = P.width >> (comp ? 1 : 0), h = P.height >> (comp ? 1 : 0);
  SaoPrm r = {0xffu, 0u, 0u};
  if (x < w && row < h) {
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(P.saoprm + ((size_t)(row >> log2ctb) * P.ctus_w + (x >> log2ctb)) * 3 + comp);
    r.w0 = ldg(pw); r.off_lo = ldg(pw + 1); r.off_hi = ldg(pw + 2);
  }
  return r;
}

// returns the group's eight output samples (the caller stores them: luma as they are, chroma paired with the other component's)
template <int W, bool NF>
__device__ inline u32x4 sao_group(const PicDev& P, int comp, const int16_t (*t)[W], int ox, int oy, int x, int row, const SaoPrm& prm) {
  const int cs = comp ? 1 : 0;
  const int w = P.width >> cs, h = P.height >> cs;
  const int log2ctb = P.log2ctu - cs;
  const int cx = x >> log2ctb, cy = row >> log2ctb;
  const uint32_t w0 = prm.w0;
  const int type = (int)(int8_t)(w0 & 0xff);
  const int16_t* line = &t[row - oy][x - ox];
  const u32x4 cur = *reinterpret_cast<const u32x4*>(line);
  if (type < 0) return cur;
  const uint32_t off_lo = prm.off_lo, off_hi = prm.off_hi;
  const int bd = P.bd[comp], maxv = (1 << bd) - 1;
  uint32_t out[4];
  if (type == HMGPU_SAO_BO) {
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
    const int ctb = 1 << log2ctb;
    const int x0 = cx << log2ctb, y0 = cy << log2ctb;
    const int x1 = min(x0 + ctb, w) - 1, y1 = min(y0 + ctb, h) - 1;
    const unsigned av = w0 >> 16;
    // neighbour rows straight from the copy (its halo holds the deblocked samples around the tile; positions outside the
    // picture hold margin samples, which the availability mask never lets through)
    auto run = [&](auto dxc, auto dyc) {
      constexpr int DX = decltype(dxc)::value, DY = decltype(dyc)::value;
      const int16_t* ra = line + DY * W;
      const int16_t* rb = line - DY * W;
      const u32x4 ea = *reinterpret_cast<const u32x4*>(ra), eb = *reinterpret_cast<const u32x4*>(rb);
      uint32_t na[4], nb[4];
      uint32_t la = 0, raa = 0, lb = 0, rbb = 0;
      if constexpr (DX < 0) { la = (uint16_t)ra[-1]; rbb = (uint16_t)rb[8]; }
      if constexpr (DX > 0) { raa = (uint16_t)ra[8]; lb = (uint16_t)rb[-1]; }
      shifted<DX>(ea, la, raa, na);
      shifted<-DX>(eb, lb, rbb, nb);
      sao_eo_core<DX, DY>(x, row, cur, na, nb, off_lo, off_hi, av, x0, y0, x1, y1, maxv, out);
    };
    switch (type) {
      case HMGPU_SAO_EO_0:   run(std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{}); break;
      case HMGPU_SAO_EO_90:  run(std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{}); break;
      case HMGPU_SAO_EO_135: run(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{}); break;
      default:               run(std::integral_constant<int, 1>{}, std::integral_constant<int, -1>{}); break;
    }
  }
  if (NF && P.any_nofilt) {
    uint32_t m[4];
    sao_exempt_mask(P, comp, x, row, m);
    const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = (c[j] & m[j]) | (out[j] & ~m[j]);
  }
  return u32x4{out[0], out[1], out[2], out[3]};
}

// the EdgeRec unit (k_prep: Bs, mean QP, exemptions) of the edge unit of direction DIR whose Q block is at luma (x, y); requested at
// kernel entry, long before the unit is filtered (its latency hides behind the tile load)
template <int DIR>
__device__ inline uint32_t edge_fetch(const PicDev& P, int x, int y, bool mine) {
  if (!(mine && x >= 0 && y >= 0 && x < P.width && y < P.height)) return 0u;
  const uint16_t* rec = reinterpret_cast<const uint16_t*>(P.edges + (size_t)(y >> 3) * (P.grid_w >> 1) + (x >> 3));
  return ldg(rec + (DIR == 0 ? ((y >> 2) & 1) : 2 + ((x >> 2) & 1)));
}

// Deblocking runs in two steps so that lanes are spent on real work only: every edge unit of the tile is CLASSIFIED by its
// own thread (edge flags, Bs; most units of large blocks stop here), the units that filter are compacted into a list in LDS
// and APPLIED by the first threads of the block -- a tile of 32x32 CUs keeps one wave busy with the filter arithmetic instead
// of three.  A list entry: unit index | Bs << 8 | P side unfiltered << 10 | Q side unfiltered << 11 | (QP + 64) << 12 |
// Q side's slice << 20.
struct SliceLf { int tc_off, beta_off, cb_off, cr_off; };     // the deblocking constants of one slice
__device__ inline SliceLf slice_lf(const SliceDev* s) {
  return {ldg(&s->tc_offset_div2), ldg(&s->beta_offset_div2), ldg(&s->pps_cb_qp_offset), ldg(&s->pps_cr_qp_offset)};
}
static_assert(HMGPU_MAX_SLICES <= 4096, "slice index must fit the 12 bits of a list entry");

// unit u of the tile with EdgeRec unit `rec` (0: not filtered) at luma (x, y) -> list entry
template <bool NF>
__device__ inline uint32_t edge_classify(const PicDev& P, uint32_t rec, int u, int x, int y) {
  if ((rec & 3u) == 0u) return 0u;
  const uint32_t bs = rec & 3u, qp = (rec >> 2) & 127u;                 // QP + 32
  const uint32_t p_nf = NF ? (rec >> 9) & 1u : 0u, q_nf = NF ? (rec >> 10) & 1u : 0u;
  // the slice whose deblocking constants apply: the one of the Q side's CTU (TComLoopFilter.cpp:565-566)
  const uint32_t slice = P.slice_idx ? (uint32_t)ldg(P.slice_idx + (size_t)(y >> P.log2ctu) * P.ctus_w + (x >> P.log2ctu)) : 0u;
  return (uint32_t)u | (bs << 8) | (p_nf << 10) | (q_nf << 11) | ((qp + 32u) << 12) | (slice << 20);
}

// append the wave's active units to its list; called by the three waves that classify
__device__ inline void push_units(uint32_t (&list)[3][64], int32_t (&count)[4], uint32_t rec, int t) {
  const unsigned long long m = __ballot(rec != 0u);
  const int wave = t >> 6, lane = t & 63;
  if (rec != 0u) list[wave][__popcll(m & ((1ull << lane) - 1ull))] = rec;
  if (lane == 0) count[wave] = __popcll(m);
}
__device__ inline uint32_t pop_unit(const uint32_t (&list)[3][64], const int32_t (&count)[4], int t) {
  const int c0 = count[0], c1 = count[1], c2 = count[2];
  if (t < c0) return list[0][t];
  if (t < c0 + c1) return list[1][t - c0];
  if (t < c0 + c1 + c2) return list[2][t - c0 - c1];
  return 0u;
}

// one classified edge unit of direction DIR, filtered on the copies
template <int DIR, bool NF>
__device__ inline void edge_apply(const PicDev& P, FilterLds& L, int x0, int y0, uint32_t rec, const SliceLf& s0) {
  if (rec == 0u) return;
  const int u = rec & 0xff, bs = (rec >> 8) & 3, qp = (int)((rec >> 12) & 0xff) - 64, slice = rec >> 20;
  const bool p_nf = NF && ((rec >> 10) & 1), q_nf = NF && ((rec >> 11) & 1);
  const int x = DIR == 0 ? x0 + 8 * (u % 9) : x0 - 4 + 4 * (u % 18);
  const int y = DIR == 0 ? y0 - 4 + 4 * (u / 9) : y0 + 8 * (u / 18);
  // offsets come from the Q side's slice (TComLoopFilter.cpp:565-566); slice 0's were fetched at kernel entry
  const SliceLf sl = slice == 0 ? s0 : slice_lf(P.slices + slice);
  const int tc_off = sl.tc_off, beta_off = sl.beta_off;
  int16_t* base = &L.y[y - (y0 - 4)][x - (x0 - 8)];
  // the unit as line pairs (filter_core.h): a = lines 0|1, b = lines 2|3, index = position across the edge
  uint32_t a[8], b[8];
  if (DIR == 0) {
    // a line = 8 contiguous samples starting 4 before the edge: 8-byte aligned in the copy
    uint32_t r[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x2 lo = *reinterpret_cast<const u32x2*>(base + i * kYW - 4), hi = *reinterpret_cast<const u32x2*>(base + i * kYW);
      r[i][0] = lo.x; r[i][1] = lo.y; r[i][2] = hi.x; r[i][3] = hi.y;
    }
    rows_to_pairs(r[0], r[1], a);
    rows_to_pairs(r[2], r[3], b);
  } else {
    // position k across the edge = 4 contiguous samples (the four lines): one 8-byte access
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(base + (k - 4) * kYW);
      a[k] = v.x; b[k] = v.y;
    }
  }
  filter_luma_unit(a, b, bs, qp, tc_off, beta_off, P.bd[0], p_nf, q_nf);
  if (DIR == 0) {
    uint32_t r[4][4];
    pairs_to_rows(a, r[0], r[1]);
    pairs_to_rows(b, r[2], r[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *reinterpret_cast<u32x2*>(base + i * kYW - 4) = (u32x2){r[i][0], r[i][1]};
      *reinterpret_cast<u32x2*>(base + i * kYW) = (u32x2){r[i][2], r[i][3]};
    }
  } else {
#pragma unroll
    for (int k = 1; k < 7; k++) *reinterpret_cast<u32x2*>(base + (k - 4) * kYW) = (u32x2){a[k], b[k]};
  }
  // chroma: Bs 2 only, edges on the 8-sample chroma grid (TComLoopFilter.cpp:225-229, 684-692, 727)
  if (bs == 2 && ((DIR == 0 ? x : y) & 15) == 0) {
    const int maxc = (1 << P.bd[1]) - 1;
#pragma unroll
    for (int comp = 1; comp < 3; comp++) {
      const int tc = chroma_tc(qp, comp == 1 ? sl.cb_off : sl.cr_off, tc_off, P.bd[comp]);
      int16_t* cb = &L.c[comp - 1][(y >> 1) - ((y0 >> 1) - 2)][(x >> 1) - ((x0 >> 1) - 8)];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int16_t* s = DIR == 0 ? cb + i * kCW : cb + i;
        const int o = DIR == 0 ? 1 : kCW;
        const int m2 = (uint16_t)s[-2 * o], m3 = (uint16_t)s[-o], m4 = (uint16_t)s[0], m5 = (uint16_t)s[o];
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        if (!p_nf) s[-o] = (int16_t)clip3(0, maxc, m3 + delta);              // xPelFilterChroma :883-890
        if (!q_nf) s[0] = (int16_t)clip3(0, maxc, m4 - delta);
      }
    }
  }
}

}  // namespace

// NF: the variant for batches in which some picture holds lossless / unfiltered PCM CUs (chosen on the host)
template <bool NF>
__global__ void __launch_bounds__(256) k_filter_fused(const PicDev* __restrict__ pics, Batch b, int tiles_x, int tiles) {
  __shared__ FilterLds L;
  // XCD-aware tile order (xcd_remap): the tiles of one picture (band) run on one XCD in raster order, so that the halo a tile
  // shares with its neighbours -- whole 128-byte lines to the left and right, rows above and below -- is found in that XCD's
  // L2 instead of being fetched from memory once per neighbour (measured: 0.9 GB of reads per batch without)
  int slot, lb;
  if (!xcd_remap(blockIdx.x, b.n, tiles, slot, lb)) return;
  const PicDev& P = pics[b.pic[slot]];
  const int x0 = (lb % tiles_x) * kTW, y0 = (lb / tiles_x) * kTH;
  const int t = threadIdx.x;
  // edge units of this thread: vertical edges x0, x0+8 .. x0+64 over the rows [y0-4, y0+68) (9 edges x 18 units), horizontal
  // edges y0 .. y0+64 over the columns [x0-4, x0+68)
  const int vx = x0 + 8 * (t % 9), vy = y0 - 4 + 4 * (t / 9), hx = x0 - 4 + 4 * (t % 18), hy = y0 + 8 * (t / 18);
  const uint32_t ev = edge_fetch<0>(P, vx, vy, t < kUnits), eh = edge_fetch<1>(P, hx, hy, t < kUnits);
  const SliceLf s0 = slice_lf(P.slices);
  // SAO groups of this thread (8 samples each): two of luma (64 rows x 8 groups), one of Cb or Cr (32 rows x 4 groups each);
  // their parameters are requested now as well
  int lx[2], ly[2];
  SaoPrm sl[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int g = t + 256 * k;
    lx[k] = x0 + (g & 7) * 8; ly[k] = y0 + (g >> 3);
    sl[k] = sao_fetch(P, 0, lx[k], ly[k]);
  }
  // (neighbouring lanes hold Cb and Cr of the same group: the plane wants them pair by pair, hmgpu_dev.h "chroma planes")
  const int ccomp = 1 + (t & 1), ccx = (x0 >> 1) + ((t >> 1) & 3) * 8, ccy = (y0 >> 1) + (t >> 3);
  const SaoPrm sc = sao_fetch(P, ccomp, ccx, ccy);
  // ---- 1. the tile and its halo, before any filtering (pictures carry margins: every address is inside the allocation):
  // all loads of the thread are issued, the edge units are classified while they fly, then the copy is written
  constexpr int VPR = kYC / 8, VPC = kCC / 8, NY = (kYH * VPR + 255) / 256, NC = (2 * kCH * VPC + 255) / 256;
  u32x4 ty[NY], tc[NC];
  {
    const int16_t* src = P.rec[0] + (ptrdiff_t)(y0 - 4) * P.pitch[0] + (x0 - 8);
#pragma unroll
    for (int k = 0; k < NY; k++) {
      const int i = t + 256 * k, r = i / VPR, v = i % VPR;
      if (i < kYH * VPR) ty[k] = ldg4(src + (ptrdiff_t)r * P.pitch[0] + 8 * v);
    }
#pragma unroll
    for (int k = 0; k < NC; k++) {
      // a vector = the (Cb, Cr) pairs of four positions of a row of the copy
      const int i = t + 256 * k, r = i / (2 * VPC), v = i % (2 * VPC);
      if (i < 2 * kCH * VPC) tc[k] = ldg4(P.rec[1] + (ptrdiff_t)((y0 >> 1) - 2 + r) * P.pitch[1] + kCStep * (((x0 >> 1) - 8) + 4 * v));
    }
  }
  if (t < 192) {
    push_units(L.unit[0], L.units[0], edge_classify<NF>(P, ev, t, vx, vy), t);
    push_units(L.unit[1], L.units[1], edge_classify<NF>(P, eh, t, hx, hy), t);
  }
#pragma unroll
  for (int k = 0; k < NY; k++) {
    const int i = t + 256 * k, r = i / VPR, v = i % VPR;
    if (i < kYH * VPR) *reinterpret_cast<u32x4*>(&L.y[r][8 * v]) = ty[k];
  }
#pragma unroll
  for (int k = 0; k < NC; k++) {
    const int i = t + 256 * k, r = i / (2 * VPC), v = i % (2 * VPC);
    if (i < 2 * kCH * VPC) {
      // the copies in LDS are one per component: the filters run on them as they did on separate planes
      *reinterpret_cast<u32x2*>(&L.c[0][r][4 * v]) = u32x2{__builtin_amdgcn_perm(tc[k].y, tc[k].x, 0x05040100u), __builtin_amdgcn_perm(tc[k].w, tc[k].z, 0x05040100u)};
      *reinterpret_cast<u32x2*>(&L.c[1][r][4 * v]) = u32x2{__builtin_amdgcn_perm(tc[k].y, tc[k].x, 0x07060302u), __builtin_amdgcn_perm(tc[k].w, tc[k].z, 0x07060302u)};
    }
  }
  __syncthreads();
  // ---- 2. vertical edges
  edge_apply<0, NF>(P, L, x0, y0, pop_unit(L.unit[0], L.units[0], t), s0);
  __syncthreads();
  // ---- 3. horizontal edges
  edge_apply<1, NF>(P, L, x0, y0, pop_unit(L.unit[1], L.units[1], t), s0);
  __syncthreads();
  // ---- 4. SAO of the tile from the deblocked copy
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (lx[k] < P.width && ly[k] < P.height)
      stg4(P.sao[0] + (size_t)ly[k] * P.pitch[0] + lx[k], sao_group<kYW, NF>(P, 0, L.y, x0 - 8, y0 - 4, lx[k], ly[k], sl[k]));
  if (ccx < (P.width >> 1) && ccy < (P.height >> 1)) {
    const u32x4 own = sao_group<kCW, NF>(P, ccomp, L.c[ccomp - 1], (x0 >> 1) - 8, (y0 >> 1) - 2, ccx, ccy, sc);
    // the even lane (Cb) writes positions 0..3 of the group, the odd lane (Cr) positions 4..7: each hands the other the half it does not write
    const bool odd = t & 1;
    const uint32_t r0 = (uint32_t)__shfl_xor((int)(odd ? own.x : own.z), 1, 64), r1 = (uint32_t)__shfl_xor((int)(odd ? own.y : own.w), 1, 64);
    const uint32_t cb0 = odd ? r0 : own.x, cb1 = odd ? r1 : own.y, cr0 = odd ? own.z : r0, cr1 = odd ? own.w : r1;
    const u32x4 o = {__builtin_amdgcn_perm(cr0, cb0, 0x05040100u), __builtin_amdgcn_perm(cr0, cb0, 0x07060302u),
                     __builtin_amdgcn_perm(cr1, cb1, 0x05040100u), __builtin_amdgcn_perm(cr1, cb1, 0x07060302u)};
    stg4(P.sao[1] + (size_t)ccy * P.pitch[1] + kCStep * (ccx + (odd ? 4 : 0)), o);
  }
}

void launch_filter_fused(const PicDev* pics, const Batch& b, int width, int height, bool nofilt, hipStream_t s) {
  const int tiles_x = (width + kTW - 1) / kTW, tiles = tiles_x * ((height + kTH - 1) / kTH);
  dim3 grid((unsigned)xcd_grid(b.n, tiles));
  if (nofilt) hipLaunchKernelGGL(k_filter_fused<true>, grid, dim3(256), 0, s, pics, b, tiles_x, tiles);
  else hipLaunchKernelGGL(k_filter_fused<false>, grid, dim3(256), 0, s, pics, b, tiles_x, tiles);
}

}  // namespace hmgpu
