// k_mc.hip -- motion compensation (TComPrediction::motionCompensation -> xPredInterBlk -> TComInterpolationFilter,
// TComYuv::addAvg; TComPrediction.cpp:514-714, TComInterpolationFilter.cpp:166-251, TComYuv.cpp:336-391).
//
// Grid-driven: the prediction of a sample depends only on where it is and on the motion stored for its 4x4 block
// (BlkInfo), never on the PU it belongs to.  One thread produces an 8x4 luma tile (two BlkInfo cells) or a 4x4 chroma
// tile of both chroma planes (one 8x8 luma area = four cells), straight from global memory through L1/L2: adjacent
// lanes read overlapping, contiguous row segments of the reference picture and write 16-byte row segments of the
// prediction.  Reference samples outside the picture are produced by coordinate clamping, which is what HM's
// border extension (TComPicYuv::extendPicBorder) amounts to.
//
// One code path for all 16 (luma) / 64 (chroma) fractional positions: always a horizontal pass to a 14-bit
// intermediate followed by a vertical pass, with the phase-0 filter {0,0,0,64,0,0,0,0}.  This is exact, not an
// approximation: HM's single-pass cases are the two-pass formula with one pass being a multiplication by 64 whose
// shift commutes with the floor (derivation in DESIGN.md "MC arithmetic"); it removes all phase-dependent branching
// from the wave.
#include "hmgpu_dev.h"
#include <algorithm>
#include <cstdlib>

namespace hmgpu {

__constant__ int8_t c_luma_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0},
                                         {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int8_t c_chroma_taps[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                           {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

struct Motion { int mvx, mvy; int ref; };     // clipped MV (quarter luma samples) + device picture handle

// TComDataCU::clipMv (TComDataCU.cpp:3102-3114): clamp against the CU origin
__device__ inline void clip_mv(const PicDev& P, int cu_x, int cu_y, int& mvx, int& mvy) {
  const int ctu = 1 << P.log2ctu;
  mvx = min((P.width + 8 - cu_x - 1) << 2, max((-ctu - 8 - cu_x + 1) * 4, mvx));
  mvy = min((P.height + 8 - cu_y - 1) << 2, max((-ctu - 8 - cu_y + 1) * 4, mvy));
}

// 14-bit intermediate prediction of a W x H tile: out[y][x] = HM's "bi" output of xPredInterBlk for that sample.
// TAPS = 8 (luma, MV fraction 2 bits) or 4 (chroma 4:2:0, fraction 3 bits).  Rows stream through: each fetched row is
// filtered horizontally and immediately scattered into the vertical accumulators it contributes to, so only the
// W x H accumulators and one row live in registers.  Reference samples are valid pixels (0 .. 2^bd-1), read unsigned.
template <int TAPS, int W, int H>
__device__ inline void predict14(const int16_t* __restrict__ ref, int pitch, int pw, int ph, int x0, int y0, int mvx, int mvy,
                                 int bd, int (&out)[H][W]) {
  constexpr int FB = TAPS == 8 ? 2 : 3;
  constexpr int BEFORE = TAPS / 2 - 1;
  constexpr int ROWS = H + TAPS - 1, COLS = W + TAPS - 1;
  constexpr int LD = (COLS + 2) & ~1;                      // samples fetched per row: even count that covers an odd start
  const int xf = mvx & ((1 << FB) - 1), yf = mvy & ((1 << FB) - 1);
  const int xs = x0 + (mvx >> FB) - BEFORE, ys = y0 + (mvy >> FB) - BEFORE;
  int cx[TAPS], cy[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; k++) {
    cx[k] = TAPS == 8 ? c_luma_taps[xf][k] : c_chroma_taps[xf][k];
    cy[k] = TAPS == 8 ? c_luma_taps[yf][k] : c_chroma_taps[yf][k];
  }
  const int head = bd >= 12 ? 2 : 14 - bd;                 // max(2, IF_INTERNAL_PREC - bitDepth)
  const int sh1 = 6 - head;
  const int off1 = -(8192 << sh1);
#pragma unroll
  for (int y = 0; y < H; y++)
#pragma unroll
    for (int x = 0; x < W; x++) out[y][x] = 0;
  const int xe = xs & ~1;
  const bool inside = xs >= 0 && xs + COLS <= pw && ys >= 0 && ys + ROWS <= ph && xe + LD <= pitch;
  const int sh_odd = (xs & 1) * 16;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    int s[COLS];
    if (inside) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(ref + (size_t)(ys + r) * pitch + xe);
      uint32_t d[LD / 2 + 1];
#pragma unroll
      for (int i = 0; i < LD / 2; i++) d[i] = q[i];
      d[LD / 2] = 0;
#pragma unroll
      for (int i = 0; i < (COLS + 1) / 2; i++) d[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh_odd);   // drop one sample if the start is odd
#pragma unroll
      for (int i = 0; i < COLS; i++) s[i] = (i & 1) ? (int)(d[i / 2] >> 16) : (int)(d[i / 2] & 0xffffu);
    } else {
      const int16_t* row = ref + (size_t)clip3(0, ph - 1, ys + r) * pitch;
#pragma unroll
      for (int i = 0; i < COLS; i++) s[i] = (uint16_t)row[clip3(0, pw - 1, xs + i)];
    }
#pragma unroll
    for (int x = 0; x < W; x++) {
      int sum = 0;
#pragma unroll
      for (int k = 0; k < TAPS; k++) sum += __mul24(s[x + k], cx[k]);
      const int t = (sum + off1) >> sh1;                   // HM: filter<N,false,true,false>, stored as Pel
#pragma unroll
      for (int y = 0; y < H; y++) {
        const int k = r - y;
        if (k >= 0 && k < TAPS) out[y][x] += __mul24(t, cy[k]);
      }
    }
  }
#pragma unroll
  for (int y = 0; y < H; y++)
#pragma unroll
    for (int x = 0; x < W; x++) out[y][x] >>= 6;          // filter<N,true,false,false>
}

// uni-prediction final rounding of a 14-bit intermediate: clip((v + 8192 + 2^(head-1)) >> head)   [= HM filter isLast]
__device__ inline int finish_uni(int v, int head, int maxv) { return clip3(0, maxv, (v + 8192 + (1 << (head - 1))) >> head); }
// TComYuv::addAvg: clip((a + b + 2^head + 2*8192) >> (head+1))
__device__ inline int finish_bi(int a, int b, int head, int maxv) { return clip3(0, maxv, (a + b + (1 << head) + 16384) >> (head + 1)); }

template <int TAPS, int W, int H>
__device__ inline void predict_tile(const PicDev& P, const PlaneSet* __restrict__ finals, int comp, int x0, int y0,
                                    const BlkInfo& bi, int cu_x, int cu_y, int16_t* __restrict__ dst) {
  const int bd = P.bd[comp];
  const int head = bd >= 12 ? 2 : 14 - bd;
  const int maxv = (1 << bd) - 1;
  const int pitch = P.pitch[comp];
  const int pw = comp ? P.width >> 1 : P.width, ph = comp ? P.height >> 1 : P.height;
  int a[H][W];
  const int l0 = (bi.flags & BF_MC_L0) ? 0 : 1;
  {
    int mvx = bi.mv[l0][0], mvy = bi.mv[l0][1];
    clip_mv(P, cu_x, cu_y, mvx, mvy);
    predict14<TAPS, W, H>(finals[bi.ref[l0]].p[comp], pitch, pw, ph, x0, y0, mvx, mvy, bd, a);
  }
  if ((bi.flags & (BF_MC_L0 | BF_MC_L1)) == (BF_MC_L0 | BF_MC_L1)) {
    int b[H][W];
    int mvx = bi.mv[1][0], mvy = bi.mv[1][1];
    clip_mv(P, cu_x, cu_y, mvx, mvy);
    predict14<TAPS, W, H>(finals[bi.ref[1]].p[comp], pitch, pw, ph, x0, y0, mvx, mvy, bd, b);
#pragma unroll
    for (int y = 0; y < H; y++)
#pragma unroll
      for (int x = 0; x < W; x++) a[y][x] = finish_bi(a[y][x], b[y][x], head, maxv);
  } else {
#pragma unroll
    for (int y = 0; y < H; y++)
#pragma unroll
      for (int x = 0; x < W; x++) a[y][x] = finish_uni(a[y][x], head, maxv);
  }
#pragma unroll
  for (int y = 0; y < H; y++) {
    int16_t* row = dst + (size_t)(y0 + y) * pitch + x0;
    if (W == 8) {
      uint4 v;
      v.x = (uint32_t)a[y][0] | ((uint32_t)a[y][1] << 16); v.y = (uint32_t)a[y][2] | ((uint32_t)a[y][3] << 16);
      v.z = (uint32_t)a[y][4] | ((uint32_t)a[y][5] << 16); v.w = (uint32_t)a[y][6] | ((uint32_t)a[y][7] << 16);
      *reinterpret_cast<uint4*>(row) = v;
    } else if (W == 4) {
      uint2 v;
      v.x = (uint32_t)a[y][0] | ((uint32_t)a[y][1] << 16); v.y = (uint32_t)a[y][2] | ((uint32_t)a[y][3] << 16);
      *reinterpret_cast<uint2*>(row) = v;
    } else {
#pragma unroll
      for (int x = 0; x < W; x++) row[x] = (int16_t)a[y][x];
    }
  }
}

__device__ inline bool same_motion(const BlkInfo& a, const BlkInfo& b) {
  const uint4 ua = *reinterpret_cast<const uint4*>(&a), ub = *reinterpret_cast<const uint4*>(&b);
  // mv[2][2], ref[2] and the MC flag bits + CU size (CU origin enters clipMv)
  return ua.x == ub.x && ua.y == ub.y && (ua.z & 0xffff) == (ub.z & 0xffff) &&
         ((a.flags ^ b.flags) & (BF_VALID | BF_INTRA | BF_MC_L0 | BF_MC_L1)) == 0 && a.log2cu == b.log2cu;
}
__device__ inline bool is_inter(const BlkInfo& b) { return (b.flags & BF_VALID) && (b.flags & (BF_MC_L0 | BF_MC_L1)); }

// one 4x4 luma cell on its own (only where the two cells of an 8x4 tile do not share their motion); kept out of line so
// that the common path's register budget is not the sum of three instantiations
__device__ __attribute__((noinline)) void luma_cell(const PicDev& P, const PlaneSet* __restrict__ finals, const BlkInfo& c, int x, int y) {
  const int cs = 1 << c.log2cu;
  predict_tile<8, 4, 4>(P, finals, 0, x, y, c, x & ~(cs - 1), y & ~(cs - 1), P.rec[0]);
}

// ---- luma: one thread per 8x4 tile; a wave covers 8x8 tiles = 64x32 samples; block = 4 consecutive strips ---------------
__global__ void __launch_bounds__(256) k_mc_luma(const PicDev* __restrict__ pics, const PlaneSet* __restrict__ finals, Batch b, int nblocks) {
  int slot, lb;
  if (!xcd_remap(blockIdx.x, b.n, nblocks, slot, lb)) return;
  const PicDev& P = pics[b.pic[slot]];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ctu_sz = 1 << P.log2ctu;
  // each wave handles one 64x32 strip; strips are enumerated over the call's CTU range as (ctu, strip inside the ctu)
  const int strips_x = max(1, ctu_sz / 64), strips_y = max(1, ctu_sz / 32);
  const int strips_per_ctu = strips_x * strips_y;
  const int sid = lb * 4 + wave;
  if (sid >= b.num_ctus[slot] * strips_per_ctu) return;
  const int ctu = b.first_ctu[slot] + sid / strips_per_ctu;
  const int s_in = sid % strips_per_ctu;
  const int cx = (ctu % P.ctus_w) * ctu_sz, cy = (ctu / P.ctus_w) * ctu_sz;
  const int x0 = cx + (s_in % strips_x) * 64 + (lane & 7) * 8;
  const int y0 = cy + (s_in / strips_x) * 32 + (lane >> 3) * 4;
  if (x0 >= cx + ctu_sz || y0 >= cy + ctu_sz || x0 >= P.width || y0 >= P.height) return;
  const BlkInfo* g = P.blk + (size_t)(y0 >> 2) * P.grid_w + (x0 >> 2);
  const BlkInfo c0 = g[0], c1 = g[1];
  if (is_inter(c0) && same_motion(c0, c1)) {
    const int cs = 1 << c0.log2cu;
    predict_tile<8, 8, 4>(P, finals, 0, x0, y0, c0, x0 & ~(cs - 1), y0 & ~(cs - 1), P.rec[0]);
  } else {
    if (is_inter(c0)) luma_cell(P, finals, c0, x0, y0);
    if (is_inter(c1)) luma_cell(P, finals, c1, x0 + 4, y0);
  }
}

// 2x2 chroma samples of one 4x4 luma cell (only when the four cells of an 8x8 area do not share their motion)
__device__ inline void chroma_cell(const PicDev& P, const PlaneSet* __restrict__ finals, const BlkInfo& c, int lx, int ly) {
  if (!is_inter(c)) return;
  const int cs = 1 << c.log2cu;
  predict_tile<4, 2, 2>(P, finals, 1, lx >> 1, ly >> 1, c, lx & ~(cs - 1), ly & ~(cs - 1), P.rec[1]);
  predict_tile<4, 2, 2>(P, finals, 2, lx >> 1, ly >> 1, c, lx & ~(cs - 1), ly & ~(cs - 1), P.rec[2]);
}

// ---- chroma: one thread per 4x4 chroma tile (8x8 luma area, four cells) of BOTH planes ----------------------------------
__global__ void __launch_bounds__(256) k_mc_chroma(const PicDev* __restrict__ pics, const PlaneSet* __restrict__ finals, Batch b, int nblocks) {
  int slot, lb;
  if (!xcd_remap(blockIdx.x, b.n, nblocks, slot, lb)) return;
  const PicDev& P = pics[b.pic[slot]];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ctu_sz = 1 << P.log2ctu;
  // a wave covers 8x8 tiles = 64x64 luma samples
  const int strips_x = max(1, ctu_sz / 64), strips_y = max(1, ctu_sz / 64);
  const int strips_per_ctu = strips_x * strips_y;
  const int sid = lb * 4 + wave;
  if (sid >= b.num_ctus[slot] * strips_per_ctu) return;
  const int ctu = b.first_ctu[slot] + sid / strips_per_ctu;
  const int s_in = sid % strips_per_ctu;
  const int cx = (ctu % P.ctus_w) * ctu_sz, cy = (ctu / P.ctus_w) * ctu_sz;
  const int x0 = cx + (s_in % strips_x) * 64 + (lane & 7) * 8;     // luma coordinates of the 8x8 area
  const int y0 = cy + (s_in / strips_x) * 64 + (lane >> 3) * 8;
  if (x0 >= cx + ctu_sz || y0 >= cy + ctu_sz || x0 >= P.width || y0 >= P.height) return;
  const BlkInfo* g = P.blk + (size_t)(y0 >> 2) * P.grid_w + (x0 >> 2);
  const BlkInfo c00 = g[0], c01 = g[1], c10 = g[P.grid_w], c11 = g[P.grid_w + 1];
  if (is_inter(c00) && same_motion(c00, c01) && same_motion(c00, c10) && same_motion(c00, c11)) {
    const int cs = 1 << c00.log2cu;
    const int cux = x0 & ~(cs - 1), cuy = y0 & ~(cs - 1);
    predict_tile<4, 4, 4>(P, finals, 1, x0 >> 1, y0 >> 1, c00, cux, cuy, P.rec[1]);
    predict_tile<4, 4, 4>(P, finals, 2, x0 >> 1, y0 >> 1, c00, cux, cuy, P.rec[2]);
  } else {
    chroma_cell(P, finals, c00, x0, y0);
    chroma_cell(P, finals, c01, x0 + 4, y0);
    chroma_cell(P, finals, c10, x0, y0 + 4);
    chroma_cell(P, finals, c11, x0 + 4, y0 + 4);
  }
}

void launch_mc_luma(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, hipStream_t s) {
  const int ctu_sz = 1 << log2ctu;
  const int strips = max_ctus * std::max(1, ctu_sz / 64) * std::max(1, ctu_sz / 32);
  const int nb = (strips + 3) / 4;
  static const bool plain = getenv("HMGPU_NO_XCD_REMAP") != nullptr;
  if (plain) { Batch bb = b; bb.n = -b.n; hipLaunchKernelGGL(k_mc_luma, dim3((unsigned)(b.n * nb)), dim3(256), 0, s, pics, finals, bb, nb); return; }
  hipLaunchKernelGGL(k_mc_luma, dim3((unsigned)xcd_grid(b.n, nb)), dim3(256), 0, s, pics, finals, b, nb);
}
void launch_mc_chroma(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, hipStream_t s) {
  const int ctu_sz = 1 << log2ctu;
  const int strips = max_ctus * std::max(1, ctu_sz / 64) * std::max(1, ctu_sz / 64);
  const int nb = (strips + 3) / 4;
  hipLaunchKernelGGL(k_mc_chroma, dim3((unsigned)xcd_grid(b.n, nb)), dim3(256), 0, s, pics, finals, b, nb);
}

// ---- kernel-level seam: xPredInterBlk on a list of blocks of one plane (tests) ------------------------------------------
template <int TAPS>
__global__ void k_mc_flat(int bit_depth, const int16_t* __restrict__ ref, int ref_stride, int ref_w, int ref_h, int n,
                          const int32_t* __restrict__ blocks, const int32_t* __restrict__ out_off, int bi, int16_t* __restrict__ dst) {
  // one thread per 2x2 output patch of a block; blockIdx.x = block
  const int blk = blockIdx.x;
  if (blk >= n) return;
  const int x0 = blocks[blk * 6 + 0], y0 = blocks[blk * 6 + 1], w = blocks[blk * 6 + 2], h = blocks[blk * 6 + 3];
  const int mvx = blocks[blk * 6 + 4], mvy = blocks[blk * 6 + 5];
  const int head = bit_depth >= 12 ? 2 : 14 - bit_depth;
  const int maxv = (1 << bit_depth) - 1;
  int16_t* out = dst + out_off[blk];
  for (int p = threadIdx.x; p < (w / 2) * (h / 2); p += blockDim.x) {
    const int px = (p % (w / 2)) * 2, py = (p / (w / 2)) * 2;
    int a[2][2];
    predict14<TAPS, 2, 2>(ref, ref_stride, ref_w, ref_h, x0 + px, y0 + py, mvx, mvy, bit_depth, a);
    for (int y = 0; y < 2; y++)
      for (int x = 0; x < 2; x++)
        out[(py + y) * w + px + x] = (int16_t)(bi ? a[y][x] : finish_uni(a[y][x], head, maxv));
  }
}

void launch_mc_flat(int is_chroma, int bit_depth, const int16_t* ref, int ref_stride, int ref_w, int ref_h, int n,
                    const int32_t* blocks, const int32_t* out_off, int bi, int16_t* dst, hipStream_t s) {
  if (is_chroma)
    hipLaunchKernelGGL(k_mc_flat<4>, dim3((unsigned)n), dim3(64), 0, s, bit_depth, ref, ref_stride, ref_w, ref_h, n, blocks, out_off, bi, dst);
  else
    hipLaunchKernelGGL(k_mc_flat<8>, dim3((unsigned)n), dim3(64), 0, s, bit_depth, ref, ref_stride, ref_w, ref_h, n, blocks, out_off, bi, dst);
}

}  // namespace hmgpu
