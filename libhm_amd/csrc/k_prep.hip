// k_prep.hip -- the data-parallel "flattener": one thread per 4x4 partition of HM's per-CTU TComDataCU arrays.
//
// Replaces three serial traversals of HM with index arithmetic on the partition's own fields:
//   * TDecCu::xDecompressCU / TComTrQuant::invRecurTransformNxN / TComTU (TDecCu.cpp:373, TComTrQuant.cpp:1550,
//     TComTU.cpp:89-171): a partition is the origin of a luma TU iff it is aligned to the TU size given by
//     depth + tr_idx; chroma follows, with the 4:2:0 rule that four 4x4 luma TUs share one 4x4 chroma TU carried
//     by the first of them (TComTU.cpp:141-151; reconstruction passes bProcessLastOfLevel = false).
//   * TComPrediction::motionCompensation (TComPrediction.cpp:514-584): prediction of a sample depends only on its
//     position and on the PU's motion, which HM replicates over all partitions of the PU (TComCUMvField), so
//     the per-partition motion is all the MC kernel needs; the identical-motion collapse (:497-512) is resolved here.
//   * TComLoopFilter::xSetLoopfilterParam / xSetEdgefilterTU / xSetEdgefilterPU (TComLoopFilter.cpp:269-409):
//     whether the left/top border of a partition is a CU, TU or PU edge follows from depth, tr_idx and part_size.
#include "hmgpu_dev.h"
#include "filter_core.h"

namespace hmgpu {

__constant__ uint8_t c_chroma_scale_420[58] = {   // HM g_aucChromaScale[CHROMA_420], TComRom.cpp:503
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};

__constant__ uint8_t c_chroma422_mode[36] = {      // g_chroma422IntraAngleMappingTable, TComRom.cpp:534-536
    0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31, 36};

// QpParam (TComTrQuant.cpp:71-100); fmt: chroma_format_idc -- the 4:2:0 table, min(qPi, 51) otherwise (g_aucChromaScale, TComRom.cpp:499-506)
__device__ inline void qp_param(int qp_y, int comp, int bd, int chroma_off, int fmt, int8_t& per, int8_t& rem) {
  const int bdo = 6 * (bd - 8);
  int base;
  if (comp == 0) base = qp_y + bdo;
  else {
    base = clip3(-bdo, 57, qp_y + chroma_off);
    base = base < 0 ? base + bdo : (fmt == 1 ? c_chroma_scale_420[base] : min(base, 51)) + bdo;
  }
  per = (int8_t)(base / 6);
  rem = (int8_t)(base % 6);
}

// One thread handles the four partitions of one 8x8 luma area (z = 4q .. 4q+3: consecutive in HM's z-scan), so every
// byte array is read as one dword per thread and both MV fields as one 16-byte vector.  8x8 is the minimum CU size,
// hence depth, part_size, pred_mode, qp and tr_idx are shared by the four partitions; cbf, transform-skip and motion are
// per partition.  Each thread can emit up to six TU records in fixed slots (four 4x4 luma TUs or one larger luma TU,
// Cb, Cr).  The block reserves one contiguous range per size class in its shard of the picture's TU lists with ONE
// global atomic per class (atomics on a single word cost ~11 ns each on MI355X: per-thread or per-wave appends would
// serialise the kernel); positions inside the range come from LDS atomics.
struct Quad {
  int ctu, z0, gx0, gy0;           // CTU address, z index of partition 0, picture coordinates (partition units) of partition 0
  int log2cu, log2tu, tr, part_size, qp_cu, sidx;
  bool valid, intra;
  uint32_t cbf[3], ts[3];          // four bytes each, partition j in byte j
  uint32_t bypass;                 // m_CUTransquantBypass of the four partitions (one CU: 8x8 is the minimum CU size)
};

__device__ inline TuRec make_tu(const PicDev& P, const Quad& q, const SliceDev* sl, int gx, int gy, int comp, int flags, int xflags, uint32_t coef_off) {
  TuRec r;
  r.x4 = (uint16_t)gx; r.y4 = (uint16_t)gy;
  r.comp_flags = (uint8_t)(comp | (flags << 2));
  const int coff = comp == 1 ? ldg(&sl->cb_qp_offset) : (comp == 2 ? ldg(&sl->cr_qp_offset) : 0);
  qp_param(q.qp_cu, comp, P.bd[comp], coff, P.fmt, r.per, r.rem);
  r.xflags = (uint8_t)xflags;
  r.coef_off = coef_off;
  return r;
}

// z index of the partition at column x, row y of a CTU (inverse of zscan_x / zscan_y)
__device__ inline int z_of(int x, int y) {
  int z = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) z |= (((x >> k) & 1) << (2 * k)) | (((y >> k) & 1) << (2 * k + 1));
  return z;
}

// what the boundary strength and the filter decisions need to know about partition z of CTU `ctu`, straight from HM's arrays:
// for the few edge units whose P side lies in a CTU that another workgroup flattens (the CTU row above, the CTU left of a
// workgroup's first).  The same fields the main path derives for its own cells.
__device__ inline BlkInfo cell_from_arrays(const PicDev& P, int ctu, int z) {
  BlkInfo bi;
  bi.mv[0][0] = bi.mv[0][1] = bi.mv[1][0] = bi.mv[1][1] = 0;
  bi.ref[0] = bi.ref[1] = -1;
  bi.qp = 0; bi.flags = 0; bi.edge = 0; bi.log2cu = 3; bi.slice = 0;
  const size_t i = (size_t)ctu * P.parts + z;
  if ((int)ldg(P.part_size + i) == HMGPU_SIZE_NONE) return bi;
  const int sidx = P.slice_idx ? ldg(P.slice_idx + ctu) : 0;
  const SliceDev* sl = P.slices + sidx;
  const int tr = ldg(P.tr_idx + i);
  const bool intra = ldg(P.pred_mode + i) == HMGPU_MODE_INTRA;
  bi.flags = BF_VALID | (intra ? BF_INTRA : 0) | (((ldg(P.cbf[0] + i) >> tr) & 1) ? BF_CBFY : 0);
  if (ldg(P.bypass + i) || (P.pcm_lf_disable && ldg(P.ipcm + i))) bi.flags |= BF_NOFILT;
  bi.qp = ldg(P.qp + i);
  bi.log2cu = (uint8_t)(P.log2ctu - ldg(P.depth + i));
  bi.slice = (uint16_t)sidx;
  if (!intra) {
    const int r0 = ldg(P.ref_idx[0] + i), r1 = ldg(P.ref_idx[1] + i);
    if (r0 >= 0) { const uint32_t w = ldg(reinterpret_cast<const uint32_t*>(P.mv[0]) + i); bi.mv[0][0] = (int16_t)(w & 0xffff); bi.mv[0][1] = (int16_t)(w >> 16); bi.ref[0] = ldg(&sl->ref_pic[0][r0]); }
    if (r1 >= 0 && ldg(&sl->slice_type) == HMGPU_B_SLICE) { const uint32_t w = ldg(reinterpret_cast<const uint32_t*>(P.mv[1]) + i); bi.mv[1][0] = (int16_t)(w & 0xffff); bi.mv[1][1] = (int16_t)(w >> 16); bi.ref[1] = ldg(&sl->ref_pic[1][r1]); }
  }
  return bi;
}

// one edge unit: Bs of the two cells that face each other, their mean QP and their exemptions (EdgeRec, hmgpu_dev.h)
__device__ inline uint16_t edge_unit(const BlkInfo& p, const BlkInfo& q, bool transform_edge) {
  const int bs = boundary_strength(p, q, transform_edge);
  if (bs == 0) return 0;
  return (uint16_t)edge_unit_pack(bs, ((int)p.qp + (int)q.qp + 1) >> 1, (p.flags & BF_NOFILT) != 0, (q.flags & BF_NOFILT) != 0);
}

// FMT: chroma_format_idc of the context (1 also for monochrome): which chroma blocks a transform unit has.  4:2:0: one per component, half
// the size, four 4x4 luma blocks sharing one 4x4; 4:4:4: the luma blocks' twins; 4:2:2: two squares of half the width, one above the other
// (TComTU.cpp:89-171, TComTrQuant.cpp:1436-1462).  Slot k of a thread = block (k & 3) of component k >> 2 (4:2:0: the six slots it always had).
template <int FMT>
__global__ void __launch_bounds__(256) k_prep(const PicDev* __restrict__ pics, Batch b, int write_blk) {
  constexpr int NS = FMT == 1 ? 6 : 12;
  __shared__ uint32_t lds_cnt[4], lds_base[4], lds_stat[2];
  // the cells of the workgroup's areas, for the edge units of their right and lower neighbours (boundary strength needs both sides)
  __shared__ __attribute__((aligned(16))) u32x4 lds_cell[256 * 4];
  // ... and the P sides that lie in CTUs other workgroups flatten: the two cells above every area of a CTU's top row (128 slots cover
  // 64 CTUs of 2 areas), the two cells left of the areas in the first column of the workgroup's first CTU.  Fetched from HM's arrays at
  // the very start (cell_from_arrays), so that their latency runs beside the thread's own loads
  __shared__ __attribute__((aligned(16))) u32x4 lds_above[256], lds_left[16];
  const PicDev& P = pics[b.pic[blockIdx.z]];
  if (threadIdx.x < 4) lds_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 2) lds_stat[threadIdx.x] = 0;
  __syncthreads();
  const int parts = P.parts;
#if defined(PREP_STOP) && PREP_STOP == 1   // experiment: stop after phase n (timing of the phases by difference)
  if (parts > 0) return;
#endif
  const int gq = blockIdx.x * 256 + threadIdx.x;                 // quad index inside the call's CTU range
  const bool active = gq < b.num_ctus[blockIdx.z] * (parts >> 2);
  Quad q; q.valid = false; q.intra = false; q.log2tu = 3; q.tr = 0; q.ctu = 0; q.z0 = 0; q.gx0 = q.gy0 = 0;
  q.log2cu = 3; q.part_size = 0; q.qp_cu = 0; q.sidx = 0; q.bypass = 0;
  // (a bit mask, not an array of flags: twelve of those as a vector crash this compiler's type legaliser)
  uint32_t hasm = 0;                                             // 4:2:0 slots: luma TU of partition 0..3 (or one larger TU in slot 0), Cb, Cr
  auto set_has = [&](int k, bool v) { hasm |= (v ? 1u : 0u) << k; };
  auto has = [&](int k) { return ((hasm >> k) & 1u) != 0; };
  uint32_t cnt[3] = {0, 0, 0}, lmask = 0;                        // compact levels: coded coefficients that start in this area (intra CUs too), coded 4x4 luma TUs
  int cls[NS] = {};
  uint32_t loc[NS] = {};
  const SliceDev* sl = P.slices;
  u32x4 cells[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  write_blk |= P.any_nofilt;                                     // SAO reads the exemption flags back (sao_exempt_mask)
  if (active) {
    q.ctu = b.first_ctu[blockIdx.z] + gq / (parts >> 2);
    q.z0 = (gq % (parts >> 2)) * 4;
    const size_t idx = (size_t)q.ctu * parts + q.z0;
    const int cx = q.ctu % P.ctus_w, cy = q.ctu / P.ctus_w;
    const int x4 = zscan_x(q.z0), y4 = zscan_y(q.z0);            // partition 0 inside the CTU; partitions 1..3 are (+1,0),(0,+1),(+1,+1)
    q.gx0 = cx * P.pw + x4; q.gy0 = cy * P.pw + y4;
    {
      // the cells this CTU's edge units need from CTUs that other workgroups flatten: the pw cells above the CTU (if the CTU above is not
      // in this workgroup) and the pw cells left of it (first CTU of the workgroup), one cell per lane of the CTU's first lanes
      const int qpc = parts >> 2, lane_c = q.z0 >> 2, t0 = (int)threadIdx.x - lane_c, cl = t0 / qpc, pw = P.pw;
      const bool need_above = cy > 0 && t0 - P.ctus_w * qpc < 0, need_left = cx > 0 && t0 - qpc < 0;
#pragma unroll 1
      for (int r = lane_c; r < 2 * pw; r += qpc) {
        const bool left = r >= pw;
        const int i = left ? r - pw : r;
#if defined(PREP_EXP) && (PREP_EXP & 1)     // experiment: no cells from other workgroups' CTUs
        if (false) {
#else
        if (left ? need_left : need_above) {
#endif
          const u32x4 v = __builtin_bit_cast(u32x4, cell_from_arrays(P, left ? q.ctu - 1 : q.ctu - P.ctus_w, left ? z_of(pw - 1, i) : z_of(i, pw - 1)));
          if (left) lds_left[i] = v; else lds_above[cl * pw + i] = v;
        }
      }
    }
#if defined(PREP_STOP) && PREP_STOP == 2
    if (parts > 0) return;
#endif
    // ---- the quad's share of HM's arrays
    const uint32_t part4 = ldg(reinterpret_cast<const uint32_t*>(P.part_size + idx));
    const uint32_t depth4 = ldg(reinterpret_cast<const uint32_t*>(P.depth + idx));
    const uint32_t pred4 = ldg(reinterpret_cast<const uint32_t*>(P.pred_mode + idx));
    const uint32_t qp4 = ldg(reinterpret_cast<const uint32_t*>(P.qp + idx));
    const uint32_t tr4 = ldg(reinterpret_cast<const uint32_t*>(P.tr_idx + idx));
    const uint32_t byp4 = ldg(reinterpret_cast<const uint32_t*>(P.bypass + idx)), pcm4 = ldg(reinterpret_cast<const uint32_t*>(P.ipcm + idx));
    const uint32_t r04 = ldg(reinterpret_cast<const uint32_t*>(P.ref_idx[0] + idx)), r14 = ldg(reinterpret_cast<const uint32_t*>(P.ref_idx[1] + idx));
    const u32x4 mv0 = ldg4(P.mv[0] + idx * 2), mv1 = ldg4(P.mv[1] + idx * 2);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      q.cbf[c] = ldg(reinterpret_cast<const uint32_t*>(P.cbf[c] + idx));
      q.ts[c] = P.tskip[c] ? ldg(reinterpret_cast<const uint32_t*>(P.tskip[c] + idx)) : 0u;
    }
    q.bypass = byp4;
    q.sidx = P.slice_idx ? ldg(P.slice_idx + q.ctu) : 0;
    sl = P.slices + q.sidx;
    q.part_size = (int)(int8_t)(part4 & 0xff);
    const int px0 = q.gx0 * 4, py0 = q.gy0 * 4;
    q.valid = px0 < P.width && py0 < P.height && q.part_size != HMGPU_SIZE_NONE;   // width/height are multiples of 8
    const int depth = depth4 & 0xff;
    q.tr = tr4 & 0xff;
    q.log2cu = P.log2ctu - depth;
    q.log2tu = q.log2cu - q.tr;
    q.intra = (pred4 & 0xff) == HMGPU_MODE_INTRA;
    const int cu_parts = 1 << (q.log2cu - 2);
    q.qp_cu = (int)ldg(P.qp + (size_t)q.ctu * parts + (q.z0 & ~(cu_parts * cu_parts - 1)));   // cu.getQP(0): first partition of the CU
    const int tu_parts = q.log2tu > 2 ? 1 << (q.log2tu - 2) : 1;
    const bool deblock = q.valid && !ldg(&sl->deblocking_disable);
    const int lf_across_slices = ldg(&sl->lf_across_slices);
    const int slice_type = ldg(&sl->slice_type);
    const bool wp = ldg(&sl->weighted_pred) != 0;
#if defined(PREP_STOP) && PREP_STOP == 3
    if (parts > 0) {
      if ((part4 ^ depth4 ^ pred4 ^ qp4 ^ tr4 ^ byp4 ^ pcm4 ^ r04 ^ r14 ^ mv0.x ^ mv1.w ^ q.cbf[0] ^ q.cbf[1] ^ q.cbf[2] ^ q.ts[0] ^ (uint32_t)q.qp_cu ^ (uint32_t)deblock ^
           (uint32_t)lf_across_slices ^ (uint32_t)slice_type ^ (uint32_t)wp) == 0x12345677u) stg(P.tu_count, 1u);
      return;
    }
#endif
    const uint32_t mvw0[4] = {mv0.x, mv0.y, mv0.z, mv0.w}, mvw1[4] = {mv1.x, mv1.y, mv1.z, mv1.w};
    // the tile's motion for k_mc.hip: what the four cells must agree on (TileMv)
    uint32_t tm_mv[4][2], tm_key[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int xj = x4 + (j & 1), yj = y4 + (j >> 1);
      const int gx = q.gx0 + (j & 1), gy = q.gy0 + (j >> 1);
      BlkInfo bi;
      bi.mv[0][0] = bi.mv[0][1] = bi.mv[1][0] = bi.mv[1][1] = 0;
      bi.ref[0] = bi.ref[1] = -1;
      bi.qp = 0; bi.flags = 0; bi.edge = 0; bi.log2cu = 3; bi.slice = 0;
      if (q.valid) {
        const int cbf_y = (q.cbf[0] >> (8 * j)) & 0xff;
        bi.flags = BF_VALID | (q.intra ? BF_INTRA : 0) | (((cbf_y >> q.tr) & 1) ? BF_CBFY : 0);
        // lossless CUs and (with pcm_loop_filter_disabled) PCM CUs are exempt from the loop filters (TComLoopFilter.cpp:629-634)
        if (((byp4 >> (8 * j)) & 0xff) || (P.pcm_lf_disable && ((pcm4 >> (8 * j)) & 0xff))) bi.flags |= BF_NOFILT;
        bi.qp = (int8_t)((qp4 >> (8 * j)) & 0xff);
        bi.log2cu = (uint8_t)q.log2cu;
        bi.slice = (uint16_t)q.sidx;
        if (!q.intra) {
          const int r0 = (int)(int8_t)((r04 >> (8 * j)) & 0xff), r1 = (int)(int8_t)((r14 >> (8 * j)) & 0xff);
          int use0 = r0 >= 0, use1 = r1 >= 0 && slice_type == HMGPU_B_SLICE;   // (a P slice has no list 1: HM leaves its indices at -1)
          if (use0) { bi.mv[0][0] = (int16_t)(mvw0[j] & 0xffff); bi.mv[0][1] = (int16_t)(mvw0[j] >> 16); bi.ref[0] = ldg(&sl->ref_pic[0][r0]); }
          if (use1) { bi.mv[1][0] = (int16_t)(mvw1[j] & 0xffff); bi.mv[1][1] = (int16_t)(mvw1[j] >> 16); bi.ref[1] = ldg(&sl->ref_pic[1][r1]); }
          // xCheckIdenticalMotion (TComPrediction.cpp:497-512): B slice, both lists, same POC and same MV -> list 0 only
          // (not with weighted bi-prediction: the two lists may carry different weights, :499)
          if (slice_type == HMGPU_B_SLICE && !wp && use0 && use1 && ldg(&sl->ref_poc[0][r0]) == ldg(&sl->ref_poc[1][r1]) && mvw0[j] == mvw1[j]) use1 = 0;
          bi.flags |= (use0 ? BF_MC_L0 : 0) | (use1 ? BF_MC_L1 : 0);
          tm_mv[j][0] = use0 ? mvw0[j] : 0u; tm_mv[j][1] = use1 ? mvw1[j] : 0u;
          // reference pictures and lists; with explicit weighted prediction the reference INDICES too (two indices may name one
          // picture with different weights)
          tm_key[j] = (uint32_t)(uint8_t)bi.ref[0] | ((uint32_t)(uint8_t)bi.ref[1] << 8) | (use0 ? 1u << 16 : 0u) | (use1 ? 1u << 17 : 0u) |
                      (wp ? ((uint32_t)(use0 ? r0 & 15 : 0) << 20) | ((uint32_t)(use1 ? r1 & 15 : 0) << 24) : 0u);
        } else { tm_mv[j][0] = tm_mv[j][1] = 0; tm_key[j] = 0; }
        // ---- deblocking edge flags (TComLoopFilter.cpp:269-409); only partitions on the 8x8 grid carry an edge
        if (deblock) {
          const int cux = xj & ~(cu_parts - 1), cuy = yj & ~(cu_parts - 1);
          const int rx = xj - cux, ry = yj - cuy;
          if ((j & 1) == 0) {                       // vertical edge at the left border (x multiple of 8)
            bool filt, trans;
            if (rx == 0) {                          // CU border: m_stLFCUParam.bLeftEdge
              bool avail = gx != 0;
              if (avail && xj == 0) {               // crosses into the left CTU: getPULeft slice/tile restrictions
                const int n = q.ctu - 1;
                if (!lf_across_slices && P.slice_idx && ldg(P.slice_idx + n) != q.sidx) avail = false;
                if (!P.lf_across_tiles && P.tile_idx && ldg(P.tile_idx + n) != ldg(P.tile_idx + q.ctu)) avail = false;
              }
              filt = trans = avail;
            } else {
              trans = xj == (xj & ~(tu_parts - 1));
              bool pu = false;                      // PU border inside the CU (xSetEdgefilterPU)
              switch (q.part_size) {
                case HMGPU_SIZE_Nx2N: case HMGPU_SIZE_NxN: pu = rx == (cu_parts >> 1); break;
                case HMGPU_SIZE_nLx2N: pu = rx == (cu_parts >> 2); break;
                case HMGPU_SIZE_nRx2N: pu = rx == cu_parts - (cu_parts >> 2); break;
                default: break;
              }
              filt = trans || pu;
            }
            bi.edge |= (filt ? BE_VER_FILTER : 0) | (trans ? BE_VER_TRANSFORM : 0);
          }
          if ((j >> 1) == 0) {                      // horizontal edge at the top border (y multiple of 8)
            bool filt, trans;
            if (ry == 0) {
              bool avail = gy != 0;
              if (avail && yj == 0) {
                const int n = q.ctu - P.ctus_w;
                if (!lf_across_slices && P.slice_idx && ldg(P.slice_idx + n) != q.sidx) avail = false;
                if (!P.lf_across_tiles && P.tile_idx && ldg(P.tile_idx + n) != ldg(P.tile_idx + q.ctu)) avail = false;
              }
              filt = trans = avail;
            } else {
              trans = yj == (yj & ~(tu_parts - 1));
              bool pu = false;
              switch (q.part_size) {
                case HMGPU_SIZE_2NxN: case HMGPU_SIZE_NxN: pu = ry == (cu_parts >> 1); break;
                case HMGPU_SIZE_2NxnU: pu = ry == (cu_parts >> 2); break;
                case HMGPU_SIZE_2NxnD: pu = ry == cu_parts - (cu_parts >> 2); break;
                default: break;
              }
              filt = trans || pu;
            }
            bi.edge |= (filt ? BE_HOR_FILTER : 0) | (trans ? BE_HOR_TRANSFORM : 0);
          }
        }
      }
      cells[j] = __builtin_bit_cast(u32x4, bi);
      if (write_blk) stg4(&P.blk[(size_t)gy * P.grid_w + gx], cells[j]);
      if (!q.valid) { tm_mv[j][0] = tm_mv[j][1] = 0; tm_key[j] = 0; }
    }
    {
      TileMv tm;
      tm.ix0 = tm.iy0 = tm.ix1 = tm.iy1 = 0; tm.frac = 0; tm.ref0 = tm.ref1 = 0; tm.flags = 0; tm.ridx = 0; tm.rmask = 0; tm.slice = 0;
      bool uni = q.valid && !q.intra && (tm_key[0] & (3u << 16)) != 0;
#pragma unroll
      for (int j = 1; j < 4; j++) uni = uni && tm_mv[j][0] == tm_mv[0][0] && tm_mv[j][1] == tm_mv[0][1] && tm_key[j] == tm_key[0];
      if (uni) {
        // TComDataCU::clipMv (TComDataCU.cpp:3102-3114) against the CU origin
        const int cs = 1 << q.log2cu, ctu_sz = 1 << P.log2ctu;
        const int cu_x = (q.gx0 * 4) & ~(cs - 1), cu_y = (q.gy0 * 4) & ~(cs - 1);
        const bool use0 = (tm_key[0] >> 16) & 1, both = use0 && ((tm_key[0] >> 17) & 1);
        const int r0 = (int)(int8_t)((r04) & 0xff), r1 = (int)(int8_t)((r14) & 0xff);
        int ix[2] = {0, 0}, iy[2] = {0, 0}, rf[2] = {0, 0}, ri[2] = {0, 0}, fr = 0;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          if (s2 == 1 && !both) break;
          const bool l1 = s2 == 1 || !use0;
          const uint32_t w = l1 ? tm_mv[0][1] : tm_mv[0][0];
          int mvx = (int)(int16_t)(w & 0xffff), mvy = (int)(int16_t)(w >> 16);
          mvx = min((P.width + 8 - cu_x - 1) << 2, max((-ctu_sz - 8 - cu_x + 1) * 4, mvx));
          mvy = min((P.height + 8 - cu_y - 1) << 2, max((-ctu_sz - 8 - cu_y + 1) * 4, mvy));
          ix[s2] = mvx >> 2; iy[s2] = mvy >> 2;
          fr |= ((mvx & 3) | ((mvy & 3) << 2)) << (4 * s2);
          rf[s2] = (int)((tm_key[0] >> (l1 ? 8 : 0)) & 0xff); ri[s2] = (l1 ? r1 : r0) & 0xff;
        }
        tm.ix0 = (int16_t)ix[0]; tm.iy0 = (int16_t)iy[0]; tm.ix1 = (int16_t)ix[1]; tm.iy1 = (int16_t)iy[1];
        tm.frac = (uint8_t)fr; tm.ref0 = (uint8_t)rf[0]; tm.ref1 = (uint8_t)rf[1]; tm.ridx = (uint8_t)((ri[0] & 15) | (ri[1] << 4));
        tm.flags = TM_ACTIVE | (both ? TM_BI : 0) | (use0 ? 0 : TM_FIRST_L1);
        tm.slice = (uint16_t)q.sidx;
      }
      // which parts of the tile the residual kernel will have written (the TUs that COVER the tile, wherever they start): the
      // same cbf chains that list the TUs further down
      if (q.valid && !q.intra && q.log2tu <= 5) {
        const uint32_t chain = (1u << (q.tr + 1)) - 1;
        uint32_t rm = 0;
        if (q.log2tu > 2) rm = ((q.cbf[0] & 0xff) & chain) == chain ? TR_LUMA : 0;
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) if ((((q.cbf[0] >> (8 * j)) & 0xff) & chain) == chain) rm |= 1u << j;
        }
        if (((q.cbf[1] & 0xff) & chain) == chain) rm |= TR_CB;
        if (((q.cbf[2] & 0xff) & chain) == chain) rm |= TR_CR;
        tm.rmask = (uint8_t)rm;
      }
      stg4(&P.tmv[(size_t)(q.gy0 >> 1) * (P.grid_w >> 1) + (q.gx0 >> 1)], __builtin_bit_cast(u32x4, tm));
    }
    // compact levels: the coded TUs that originate here (same structural rule for inter and intra CUs)
    if (q.valid && q.log2tu <= 5) {
      const uint32_t chain = (1u << (q.tr + 1)) - 1;
      if (q.log2tu > 2) {
        const int tu_parts2 = 1 << (q.log2tu - 2);
        if ((x4 & (tu_parts2 - 1)) == 0 && (y4 & (tu_parts2 - 1)) == 0) {
          if (((q.cbf[0] & 0xff) & chain) == chain) cnt[0] = 1u << (2 * q.log2tu);
          if (((q.cbf[1] & 0xff) & chain) == chain) cnt[1] = 1u << (2 * q.log2tu - 2);
          if (((q.cbf[2] & 0xff) & chain) == chain) cnt[2] = 1u << (2 * q.log2tu - 2);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) if ((((q.cbf[0] >> (8 * j)) & 0xff) & chain) == chain) lmask |= 1u << j;
        cnt[0] = 16u * __popc(lmask);
        if (((q.cbf[1] & 0xff) & chain) == chain) cnt[1] = 16;
        if (((q.cbf[2] & 0xff) & chain) == chain) cnt[2] = 16;
      }
    }
#if defined(PREP_STOP) && PREP_STOP == 4
    if (parts > 0) return;
#endif
    if (q.valid) atomicAdd(&lds_stat[q.intra ? 0 : 1], 4u);
    {
      // how many of the CTU's 8x8 areas are intra (k_intra: a CTU with few of them does not stage its samples).  The areas of a CTU are
      // parts / 4 = 64, 16 or 4 consecutive lanes of one wave
      const int qpc = parts >> 2, ln = threadIdx.x & 63, seg = ln & ~(qpc - 1);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(q.valid && q.intra);
      const unsigned long long sm = qpc >= 64 ? m : (m >> seg) & ((1ull << qpc) - 1ull);
      if (ln == seg && sm) stg(P.ctu_intra + q.ctu, (uint8_t)__popcll(sm));
    }
    // ---- which transform units originate in this 8x8 area.  The TUs of intra CUs are listed too (not those of PCM CUs): their residual
    // does not depend on the neighbours, k_itx computes it ahead of k_intra, which walks the TUs in dependency order and only adds it.
    // cbf bit d of a partition = cbf of its ancestor TU node at transform depth d (TComDataCU.h:310); HM descends only
    // while every node on the way has its bit set (TComTrQuant.cpp:1558-1564)
    if (q.valid && q.log2tu <= 5 && (!q.intra || (P.has_intra_dir && !(pcm4 & 0xff)))) {
      const uint32_t chain = (1u << (q.tr + 1)) - 1;
      if (q.log2tu > 2) {
        // at most one luma TU (and its chroma TUs, half the size) starts here: when partition 0 is aligned to the TU
        const int tu_parts = 1 << (q.log2tu - 2);
        if ((x4 & (tu_parts - 1)) == 0 && (y4 & (tu_parts - 1)) == 0) {
          set_has(0, ((q.cbf[0] & 0xff) & chain) == chain); cls[0] = q.log2tu - 2;
          if constexpr (FMT == 1) {
            set_has(4, ((q.cbf[1] & 0xff) & chain) == chain); cls[4] = q.log2tu - 3;
            set_has(5, ((q.cbf[2] & 0xff) & chain) == chain); cls[5] = q.log2tu - 3;
          } else {
#pragma unroll
            for (int c = 1; c < 3; c++) {
              set_has(4 * c, ((q.cbf[c] & 0xff) & chain) == chain); cls[4 * c] = FMT == 3 ? q.log2tu - 2 : q.log2tu - 3;
              if constexpr (FMT == 2) { set_has(4 * c + 1, has(4 * c)); cls[4 * c + 1] = cls[4 * c]; }      // the lower square: transformed whenever the block is
            }
          }
        }
      } else {
        // four 4x4 luma TUs; the one 4x4 chroma TU of the 8x8 node rides with the first of them (TComTU.cpp:141-151)
#pragma unroll
        for (int j = 0; j < 4; j++) { set_has(j, (((q.cbf[0] >> (8 * j)) & 0xff) & chain) == chain); cls[j] = 0; }
        if constexpr (FMT == 1) {
          set_has(4, ((q.cbf[1] & 0xff) & chain) == chain); cls[4] = 0;
          set_has(5, ((q.cbf[2] & 0xff) & chain) == chain); cls[5] = 0;
        } else if constexpr (FMT == 3) {
          // 4:4:4: every 4x4 luma block has its chroma twins
#pragma unroll
          for (int c = 1; c < 3; c++)
#pragma unroll
            for (int j = 0; j < 4; j++) { set_has(4 * c + j, (((q.cbf[c] >> (8 * j)) & 0xff) & chain) == chain); cls[4 * c + j] = 0; }
        } else {
          // 4:2:2: the 4x8 chroma block of the 8x8 node, two 4x4 squares
#pragma unroll
          for (int c = 1; c < 3; c++) { const bool coded = ((q.cbf[c] & 0xff) & chain) == chain; set_has(4 * c, coded); set_has(4 * c + 1, coded); cls[4 * c] = cls[4 * c + 1] = 0; }
        }
      }
#pragma unroll
      for (int k = 0; k < NS; k++) if (has(k)) loc[k] = atomicAdd(&lds_cnt[cls[k]], 1u);
    }
  }
  // compact levels: where this area's TUs start = the CTU's start + the coefficients of the areas before it in the CTU (z-order).
  // The areas of a CTU are consecutive lanes of one wave (64, 16 or 4 of them): an inclusive scan over the wave, minus what lies
  // before the CTU's first lane.
  uint32_t coff[3] = {0, 0, 0};
  if (P.coef_start[0] != nullptr) {
    const int lane = threadIdx.x & 63, qpc = parts >> 2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      uint32_t incl = cnt[c];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      const uint32_t excl = incl - cnt[c];
      const uint32_t seg = __shfl(excl, lane & ~(qpc - 1));
      coff[c] = excl - seg;
      if (active) {
        coff[c] += ldg(P.coef_start[c] + q.ctu);
        stg(P.quad_off[c] + (size_t)q.ctu * qpc + (q.z0 >> 2), coff[c]);
      }
    }
  }
#if defined(PREP_STOP) && PREP_STOP == 5
  if (parts > 0) return;
#endif
#if !(defined(PREP_EXP) && (PREP_EXP & 4)) // experiment: no cell exchange
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; j++) lds_cell[threadIdx.x * 4 + j] = cells[j];
  }
#endif
  __syncthreads();
  // ---- the area's four edge units (xGetBoundaryStrengthSingle, TComLoopFilter.cpp:411-537): the Q side is one of its own cells, the P side
  // the cell to the left / above -- another thread's, through LDS where that thread belongs to this workgroup (the areas of a CTU are 64,
  // 16 or 4 consecutive threads in z-order, consecutive CTUs follow each other)
  if (active) {
    EdgeRec er; er.v[0] = er.v[1] = er.h[0] = er.h[1] = 0;
#if defined(PREP_EXP) && (PREP_EXP & 2)     // experiment: no edge units
    if (false) {
#else
    if (q.valid) {
#endif
      const int x4 = zscan_x(q.z0), y4 = zscan_y(q.z0), qpc = parts >> 2;
      const int t0 = (int)threadIdx.x - (q.z0 >> 2);             // the thread of the CTU's first area
#pragma unroll
      for (int k = 0; k < 2; k++) {
        {
          const BlkInfo Q = __builtin_bit_cast(BlkInfo, cells[2 * k]);
          if (Q.edge & BE_VER_FILTER) {
            BlkInfo Pn;
            if (x4 > 0) Pn = __builtin_bit_cast(BlkInfo, lds_cell[(t0 + (z_of(x4 - 2, y4) >> 2)) * 4 + 2 * k + 1]);
            else {
              const int tn = t0 - qpc + (z_of(P.pw - 2, y4) >> 2);
              if (tn >= 0) Pn = __builtin_bit_cast(BlkInfo, lds_cell[tn * 4 + 2 * k + 1]);
              else Pn = __builtin_bit_cast(BlkInfo, lds_left[y4 + k]);
            }
            er.v[k] = edge_unit(Pn, Q, (Q.edge & BE_VER_TRANSFORM) != 0);
          }
        }
        {
          const BlkInfo Q = __builtin_bit_cast(BlkInfo, cells[k]);
          if (Q.edge & BE_HOR_FILTER) {
            BlkInfo Pn;
            if (y4 > 0) Pn = __builtin_bit_cast(BlkInfo, lds_cell[(t0 + (z_of(x4, y4 - 2) >> 2)) * 4 + 2 + k]);
            else {
              const int tn = t0 - P.ctus_w * qpc + (z_of(x4, P.pw - 2) >> 2);
              if (tn >= 0) Pn = __builtin_bit_cast(BlkInfo, lds_cell[tn * 4 + 2 + k]);
              else Pn = __builtin_bit_cast(BlkInfo, lds_above[(t0 / qpc) * P.pw + x4 + k]);
            }
            er.h[k] = edge_unit(Pn, Q, (Q.edge & BE_HOR_TRANSFORM) != 0);
          }
        }
      }
    }
    stg2(reinterpret_cast<uint32_t*>(P.edges + (size_t)(q.gy0 >> 1) * (P.grid_w >> 1) + (q.gx0 >> 1)), __builtin_bit_cast(u32x2, er));
  }
#if defined(PREP_STOP) && PREP_STOP == 6
  if (parts > 0) return;
#endif
  const int shard = blockIdx.x & (kTuShards - 1);
  if (threadIdx.x < 4) {
    const uint32_t n = lds_cnt[threadIdx.x];
    lds_base[threadIdx.x] = n ? atomicAdd(&P.tu_count[threadIdx.x * kTuShards + shard], n) : 0u;
  } else if (threadIdx.x < 6) {
    const uint32_t n = lds_stat[threadIdx.x - 4];
    if (n) atomicAdd(&P.stats[(threadIdx.x - 4) * kTuShards + shard], (unsigned long long)n);
  }
  __syncthreads();
#if defined(PREP_STOP) && PREP_STOP == 7
  if (parts > 0) return;
#endif
  const int ctu_luma = 1 << (2 * P.log2ctu);
#pragma unroll
  for (int k = 0; k < NS; k++) {
    if (!has(k)) continue;
    const int comp = FMT == 1 ? (k < 4 ? 0 : k - 3) : k >> 2;
    // the partition the block's flags are stored at (j: of this area), its position relative to the area's first partition, its place among the levels
    int j = FMT == 1 ? (k < 4 ? k : 0) : (k & 3), dx = j & 1, dy = j >> 1, bsize = comp == 0 ? q.log2tu : max(q.log2tu - 1, 2);
    int ts = (q.ts[comp] >> (8 * j)) & 0xff;
    uint32_t off = comp == 0 ? (uint32_t)q.ctu * ctu_luma + 16u * (q.z0 + j) : (uint32_t)q.ctu * (ctu_luma >> 2) + 4u * q.z0;
    if constexpr (FMT == 3) {
      bsize = q.log2tu;
      off = (uint32_t)q.ctu * ctu_luma + 16u * (q.z0 + j);
    } else if constexpr (FMT == 2) {
      if (comp) {
        // square `j` (0 upper, 1 lower) of the chroma block of the node: the lower one lies half the node's height down -- with the flags of
        // the lower half of the node's partitions -- and follows the upper one among the levels
        const int l2 = max(q.log2tu, 3) - 1, n = 1 << l2, lower = j;
        bsize = l2; dx = 0; dy = lower ? n >> 2 : 0;
        off = (uint32_t)q.ctu * (ctu_luma >> 1) + 8u * q.z0 + (lower ? (uint32_t)(n * n) : 0u);
        ts = !lower ? (q.ts[comp] & 0xff) : q.log2tu <= 3 ? (q.ts[comp] >> 16) & 0xff : (P.tskip[comp] ? (int)ldg(P.tskip[comp] + (size_t)q.ctu * P.parts + q.z0 + (1 << (2 * (q.log2tu - 2) - 1))) : 0);
        j = 0;
      }
    }
    // bit 0: DST, bit 1: transform skip, bit 2: cu_transquant_bypass, bits 3-4: RDPCM of a block that skipped the transform, bit 5: intra
    int flags = ((comp == 0 && q.log2tu == 2 && q.intra) ? 1 : 0) | ((ts & 1) ? 2 : 0) | ((q.bypass & 0xff) ? 4 : 0) | (q.intra ? 32 : 0);
    int xflags = 0;
    if ((flags & 6) && P.range_ext) {
      if (!q.intra) {
        if (P.range_ext & HMGPU_REXT_EXPLICIT_RDPCM) flags |= ((ts >> 1) & 3) << 3;          // the mode parsed with the block
      } else {
        // rotation of 4x4 blocks (TComTU::isNonTransformedResidualRotated); implicit RDPCM along the final prediction mode (invRdpcmNxN,
        // TComTrQuant.cpp:1748-1760): DM_CHROMA = the luma mode of the CU's first partition (4:4:4: of the same partition), as k_intra
        // derives it; 4:2:2: through the mode table
        if ((P.range_ext & HMGPU_REXT_ROTATION) && bsize == 2) xflags |= 1;
        if (P.range_ext & HMGPU_REXT_IMPLICIT_RDPCM) {
          const size_t cb = (size_t)q.ctu * P.parts;
          int mode = ldg(P.intra_dir[comp ? 1 : 0] + cb + q.z0 + j);
          if (comp && mode == 36) mode = ldg(P.intra_dir[0] + cb + (FMT == 3 ? q.z0 + j : (q.z0 & ~((1 << (2 * (q.log2cu - 2))) - 1))));
          if (FMT == 2 && comp) mode = c_chroma422_mode[mode];
          flags |= (mode == 10 ? 1 : mode == 26 ? 2 : 0) << 3;
        }
      }
    }
    if (FMT == 1 && P.coef_start[0] != nullptr) off = coff[comp] + (comp == 0 ? 16u * __popc(lmask & ((1u << j) - 1u)) : 0u);
    const TuRec r = make_tu(P, q, sl, q.gx0 + dx, q.gy0 + dy, comp, flags, xflags, off);
    const int c = cls[k];
    const uint32_t i = lds_base[c] + loc[k];
    if (i < P.tu_cap[c]) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(P.tu[c] + (size_t)shard * P.tu_cap[c] + i);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&r);
      stg(dst, src[0]); stg(dst + 1, src[1]); stg(dst + 2, src[2]);
    }
  }
}

// resets the TU list lengths of every picture of the batch (one launch instead of one memset per picture)
__global__ void k_zero_counts(const PicDev* __restrict__ pics, Batch b, int intra) {
  const PicDev& P = pics[b.pic[blockIdx.x]];
  if (threadIdx.x < 4 * kTuShards) stg(P.tu_count + threadIdx.x, 0u);
  // intra state of this call: per-CTU "holds intra CUs" flags of the CTU range, "done" flags of all CTUs
  for (int i = threadIdx.x; i < b.num_ctus[blockIdx.x]; i += blockDim.x) stg(P.ctu_intra + b.first_ctu[blockIdx.x] + i, (uint8_t)0);
  if (intra) for (int i = threadIdx.x; i < 3 * P.num_ctus; i += blockDim.x) stg(P.intra_done + i, 0u);
}

// write_blk: the call runs kernels that read the BlkInfo grid (the cells kernels of mixed-motion tiles); pictures with exempt CUs add
// themselves (SAO's exemption mask).  Everything else reads TileMv and EdgeRec only.
void launch_prep(const PicDev* pics, const Batch& b, int max_ctus, int parts, bool intra, bool write_blk, int fmt, hipStream_t s) {
  hipLaunchKernelGGL(k_zero_counts, dim3((unsigned)b.n), dim3(256), 0, s, pics, b, intra ? 1 : 0);
  dim3 grid((unsigned)(((size_t)max_ctus * (parts / 4) + 255) / 256), 1, (unsigned)b.n);
  if (fmt == 3) hipLaunchKernelGGL(k_prep<3>, grid, dim3(256), 0, s, pics, b, write_blk ? 1 : 0);
  else if (fmt == 2) hipLaunchKernelGGL(k_prep<2>, grid, dim3(256), 0, s, pics, b, write_blk ? 1 : 0);
  else hipLaunchKernelGGL(k_prep<1>, grid, dim3(256), 0, s, pics, b, write_blk ? 1 : 0);
}

}  // namespace hmgpu
