// k_mc_cells.hip -- motion compensation of the 4x4 cells that the picture kernels of k_mc.hip leave out, and the
// kernel-level test seam.  (TComPrediction::motionCompensation -> xPredInterBlk -> TComInterpolationFilter, TComYuv::addAvg;
// TComPrediction.cpp:514-714, TComInterpolationFilter.cpp:166-251, TComYuv.cpp:336-391.)
//
// k_mc.hip predicts every 8x8 luma tile whose four 4x4 cells share their motion.  The tiles that do not -- 8x4 / 4x8 PUs,
// the 4- and 12-sample parts of AMP in 16x16 CUs -- are predicted here, one cell per call of the register path
// (predict14: the whole window of the cell fetched into registers, H pass, V pass).  The host launches these kernels
// only for slice calls whose part_size / depth arrays contain such PUs (hmgpu_api.hip: SliceCall::cells).
#include "mc_core.h"

namespace hmgpu {


// WP: the variant for calls whose slice uses explicit weighted prediction (chosen on the host: the common kernels do not
// carry its code or its registers)
template <bool WP>
__global__ void __launch_bounds__(256) k_mc_luma_cells(const PicDev* __restrict__ pics, const PlaneSet* __restrict__ finals, Batch b, int nblocks) {
  int slot, lb, x0, y0;
  if (!xcd_remap(blockIdx.x, b.n, nblocks, slot, lb)) return;
  const PicDev& P = pics[b.pic[slot]];
  if (!tile_origin(P, b, slot, lb, x0, y0)) return;
  const BlkInfo* g = P.blk + (size_t)(y0 >> 2) * P.grid_w + (x0 >> 2);
  const BlkInfo c00 = ld_blk(g), c01 = ld_blk(g + 1), c10 = ld_blk(g + P.grid_w), c11 = ld_blk(g + P.grid_w + 1);
  const uint32_t* tmw = reinterpret_cast<const uint32_t*>(P.tmv + (size_t)(y0 >> 3) * (P.grid_w >> 1) + (x0 >> 3));
  if (ldg(tmw + 2) >> 24 & TM_ACTIVE) return;               // done by k_mc_luma
  const uint32_t rmask = (ldg(tmw + 3) >> 8) & 0xff;
  if (is_inter(c00)) luma_cell<WP>(P, finals, c00, x0, y0, rmask);
  if (is_inter(c01)) luma_cell<WP>(P, finals, c01, x0 + 4, y0, rmask);
  if (is_inter(c10)) luma_cell<WP>(P, finals, c10, x0, y0 + 4, rmask);
  if (is_inter(c11)) luma_cell<WP>(P, finals, c11, x0 + 4, y0 + 4, rmask);
}

// chroma: the 2x2 samples of BOTH chroma planes under each 4x4 luma cell
template <bool WP>
__global__ void __launch_bounds__(256) k_mc_chroma_cells(const PicDev* __restrict__ pics, const PlaneSet* __restrict__ finals, Batch b, int nblocks) {
  int slot, lb, x0, y0;
  if (!xcd_remap(blockIdx.x, b.n, nblocks, slot, lb)) return;
  const PicDev& P = pics[b.pic[slot]];
  if (!tile_origin(P, b, slot, lb, x0, y0)) return;
  const BlkInfo* g = P.blk + (size_t)(y0 >> 2) * P.grid_w + (x0 >> 2);
  const BlkInfo c00 = ld_blk(g), c01 = ld_blk(g + 1), c10 = ld_blk(g + P.grid_w), c11 = ld_blk(g + P.grid_w + 1);
  const uint32_t* tmw = reinterpret_cast<const uint32_t*>(P.tmv + (size_t)(y0 >> 3) * (P.grid_w >> 1) + (x0 >> 3));
  if (ldg(tmw + 2) >> 24 & TM_ACTIVE) return;               // done by k_mc_chroma
  const uint32_t rmask = (ldg(tmw + 3) >> 8) & 0xff;
  if (is_inter(c00)) chroma_cell<WP>(P, finals, c00, x0, y0, rmask);
  if (is_inter(c01)) chroma_cell<WP>(P, finals, c01, x0 + 4, y0, rmask);
  if (is_inter(c10)) chroma_cell<WP>(P, finals, c10, x0, y0 + 4, rmask);
  if (is_inter(c11)) chroma_cell<WP>(P, finals, c11, x0 + 4, y0 + 4, rmask);
}

// chroma of 4:2:2 / 4:4:4 pictures: EVERY inter cell of every tile through the register path (correct first: these formats have no
// LDS-staged picture kernel yet)
template <bool WP, int CSX, int CSY>
__global__ void __launch_bounds__(256) k_mc_chroma_fmt(const PicDev* __restrict__ pics, const PlaneSet* __restrict__ finals, Batch b, int nblocks) {
  int slot, lb, x0, y0;
  if (!xcd_remap(blockIdx.x, b.n, nblocks, slot, lb)) return;
  const PicDev& P = pics[b.pic[slot]];
  if (!tile_origin(P, b, slot, lb, x0, y0)) return;
  const BlkInfo* g = P.blk + (size_t)(y0 >> 2) * P.grid_w + (x0 >> 2);
  const BlkInfo c00 = ld_blk(g), c01 = ld_blk(g + 1), c10 = ld_blk(g + P.grid_w), c11 = ld_blk(g + P.grid_w + 1);
  if (is_inter(c00)) chroma_cell_fmt<WP, CSX, CSY>(P, finals, c00, x0, y0);
  if (is_inter(c01)) chroma_cell_fmt<WP, CSX, CSY>(P, finals, c01, x0 + 4, y0);
  if (is_inter(c10)) chroma_cell_fmt<WP, CSX, CSY>(P, finals, c10, x0, y0 + 4);
  if (is_inter(c11)) chroma_cell_fmt<WP, CSX, CSY>(P, finals, c11, x0 + 4, y0 + 4);
}

static int mc_blocks(int max_ctus, int log2ctu) {
  const int sq = std::max(1, (1 << log2ctu) / 64);
  return (max_ctus * sq * sq + 3) / 4;
}
void launch_mc_luma_cells(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, bool wp, hipStream_t s) {
  const int nb = mc_blocks(max_ctus, log2ctu);
  if (wp) hipLaunchKernelGGL(k_mc_luma_cells<true>, dim3((unsigned)xcd_grid(b.n, nb)), dim3(256), 0, s, pics, finals, b, nb);
  else hipLaunchKernelGGL(k_mc_luma_cells<false>, dim3((unsigned)xcd_grid(b.n, nb)), dim3(256), 0, s, pics, finals, b, nb);
}
void launch_mc_chroma_cells(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, bool wp, hipStream_t s) {
  const int nb = mc_blocks(max_ctus, log2ctu);
  if (wp) hipLaunchKernelGGL(k_mc_chroma_cells<true>, dim3((unsigned)xcd_grid(b.n, nb)), dim3(256), 0, s, pics, finals, b, nb);
  else hipLaunchKernelGGL(k_mc_chroma_cells<false>, dim3((unsigned)xcd_grid(b.n, nb)), dim3(256), 0, s, pics, finals, b, nb);
}

void launch_mc_chroma_fmt(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, int fmt, bool wp, hipStream_t s) {
  const int nb = mc_blocks(max_ctus, log2ctu);
  const dim3 grid((unsigned)xcd_grid(b.n, nb));
  if (fmt == 3) {
    if (wp) hipLaunchKernelGGL((k_mc_chroma_fmt<true, 0, 0>), grid, dim3(256), 0, s, pics, finals, b, nb);
    else hipLaunchKernelGGL((k_mc_chroma_fmt<false, 0, 0>), grid, dim3(256), 0, s, pics, finals, b, nb);
  } else {
    if (wp) hipLaunchKernelGGL((k_mc_chroma_fmt<true, 1, 0>), grid, dim3(256), 0, s, pics, finals, b, nb);
    else hipLaunchKernelGGL((k_mc_chroma_fmt<false, 1, 0>), grid, dim3(256), 0, s, pics, finals, b, nb);
  }
}

// ---- kernel-level seam: xPredInterBlk on a list of blocks of one plane (tests).  The plane carries replicated margins
// (built by the host wrapper); blocks are cut into the same 8x8 / 4x4 tiles the picture kernels use, remainders into 2x2.
template <int TAPS, int T>
__global__ void k_mc_flat(int bit_depth, const int16_t* __restrict__ ref, int ref_stride, int n, const int32_t* __restrict__ blocks,
                          const int32_t* __restrict__ out_off, int bi, int16_t* __restrict__ dst) {
  const int blk = blockIdx.x;
  if (blk >= n) return;
  const int x0 = blocks[blk * 6 + 0], y0 = blocks[blk * 6 + 1], w = blocks[blk * 6 + 2], h = blocks[blk * 6 + 3];
  const int mvx = blocks[blk * 6 + 4], mvy = blocks[blk * 6 + 5];
  const int head = bit_depth >= 12 ? 2 : 14 - bit_depth;
  const int maxv = (1 << bit_depth) - 1;
  int16_t* out = dst + out_off[blk];
  const int tw = w / T, th = h / T;                        // full T x T tiles
  for (int p = threadIdx.x; p < tw * th; p += blockDim.x) {
    const int px = (p % tw) * T, py = (p / tw) * T;
    int a[T][T];
    predict14<TAPS, T, T>(ref, ref_stride, x0 + px, y0 + py, mvx, mvy, bit_depth, a);
    for (int y = 0; y < T; y++)
      for (int x = 0; x < T; x++) out[(py + y) * w + px + x] = (int16_t)(bi ? (a[y][x] >> 6) : finish_uni(a[y][x], head, maxv));
  }
  // remainder columns / rows in 2x2 patches
  const int rw = w / 2, rh = h / 2;
  for (int p = threadIdx.x; p < rw * rh; p += blockDim.x) {
    const int px = (p % rw) * 2, py = (p / rw) * 2;
    if (px < tw * T && py < th * T) continue;
    int a[2][2];
    predict14<TAPS, 2, 2>(ref, ref_stride, x0 + px, y0 + py, mvx, mvy, bit_depth, a);
    for (int y = 0; y < 2; y++)
      for (int x = 0; x < 2; x++) out[(py + y) * w + px + x] = (int16_t)(bi ? (a[y][x] >> 6) : finish_uni(a[y][x], head, maxv));
  }
}

void launch_mc_flat(int is_chroma, int bit_depth, const int16_t* ref, int ref_stride, int ref_w, int ref_h, int n,
                    const int32_t* blocks, const int32_t* out_off, int bi, int16_t* dst, hipStream_t s) {
  (void)ref_w; (void)ref_h;
  if (is_chroma)
    hipLaunchKernelGGL((k_mc_flat<4, 4>), dim3((unsigned)n), dim3(64), 0, s, bit_depth, ref, ref_stride, n, blocks, out_off, bi, dst);
  else
    hipLaunchKernelGGL((k_mc_flat<8, 8>), dim3((unsigned)n), dim3(64), 0, s, bit_depth, ref, ref_stride, n, blocks, out_off, bi, dst);
}

}  // namespace hmgpu
