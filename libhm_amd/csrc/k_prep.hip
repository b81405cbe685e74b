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

namespace hmgpu {

__constant__ uint8_t c_chroma_scale_420[58] = {   // HM g_aucChromaScale[CHROMA_420], TComRom.cpp:503
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51};

// QpParam (TComTrQuant.cpp:71-100)
__device__ inline void qp_param(int qp_y, int comp, int bd, int chroma_off, int8_t& per, int8_t& rem) {
  const int bdo = 6 * (bd - 8);
  int base;
  if (comp == 0) base = qp_y + bdo;
  else {
    base = clip3(-bdo, 57, qp_y + chroma_off);
    base = base < 0 ? base + bdo : c_chroma_scale_420[base] + bdo;
  }
  per = (int8_t)(base / 6);
  rem = (int8_t)(base % 6);
}

// A thread stages at most three TU records (Y, Cb, Cr); the block then reserves one contiguous range per size class
// in its shard of the picture's TU lists with ONE global atomic per class (atomics on a single word cost ~11 ns each on
// MI355X: per-thread or per-wave appends would serialise the whole kernel) and writes the records there.
struct TuStage { TuRec r[3]; int cls[3]; int loc[3]; int n; };

__device__ inline void stage_tu(const PicDev& P, TuStage& st, uint32_t* lds_cnt, int log2size, int x4, int y4, int comp,
                                int flags, int qp_y, const SliceDev* sl, uint32_t coef_off) {
  TuRec r;
  r.x4 = (uint16_t)x4; r.y4 = (uint16_t)y4;
  r.comp_flags = (uint8_t)(comp | (flags << 2));
  const int coff = comp == 1 ? ldg(&sl->cb_qp_offset) : (comp == 2 ? ldg(&sl->cr_qp_offset) : 0);
  qp_param(qp_y, comp, P.bd[comp], coff, r.per, r.rem);
  r.pad = 0;
  r.coef_off = coef_off;
  const int k = st.n++;
  st.r[k] = r;
  st.cls[k] = log2size - 2;
  st.loc[k] = (int)atomicAdd(&lds_cnt[log2size - 2], 1u);      // LDS atomic: position inside the block's range
}

// one partition: BlkInfo + staged TU records + counts of intra/inter partitions
__device__ inline void prep_partition(const PicDev& P, const Batch& b, int gpart, TuStage& st, uint32_t* lds_cnt, uint32_t* lds_stat) {
  const int parts = P.parts;
  const int ctu = b.first_ctu[blockIdx.z] + gpart / parts;
  const int z = gpart % parts;
  const size_t idx = (size_t)ctu * parts + z;
  const int cx = ctu % P.ctus_w, cy = ctu / P.ctus_w;
  const int x4 = zscan_x(z), y4 = zscan_y(z);              // inside the CTU, partition units
  const int gx = cx * P.pw + x4, gy = cy * P.pw + y4;      // picture, partition units
  const int px = gx * 4, py = gy * 4;

  BlkInfo bi;
  bi.mv[0][0] = bi.mv[0][1] = bi.mv[1][0] = bi.mv[1][1] = 0;
  bi.ref[0] = bi.ref[1] = -1;
  bi.qp = 0; bi.flags = 0; bi.edge = 0; bi.log2cu = 3; bi.slice = 0;
  BlkInfo* out = &P.blk[(size_t)gy * P.grid_w + gx];

  const int part_size = ldg(P.part_size + (idx));
  if (px >= P.width || py >= P.height || part_size == HMGPU_SIZE_NONE) { stg4(out, __builtin_bit_cast(u32x4, bi)); return; }

  const int sidx = P.slice_idx ? ldg(P.slice_idx + (ctu)) : 0;
  const SliceDev* sl = P.slices + sidx;
  const int depth = ldg(P.depth + (idx));
  const int tr = ldg(P.tr_idx + (idx));
  const int log2cu = P.log2ctu - depth;
  const int log2tu = log2cu - tr;
  const int cu_parts = 1 << (log2cu - 2);                  // CU width in partitions
  const int tu_parts = log2tu > 2 ? 1 << (log2tu - 2) : 1;
  const int cux = x4 & ~(cu_parts - 1), cuy = y4 & ~(cu_parts - 1);
  const int rx = x4 - cux, ry = y4 - cuy;                  // inside the CU
  const bool intra = ldg(P.pred_mode + (idx)) == HMGPU_MODE_INTRA;
  const int cbf_y = ldg(P.cbf[0] + (idx));

  bi.flags = BF_VALID | (intra ? BF_INTRA : 0) | (((cbf_y >> tr) & 1) ? BF_CBFY : 0);
  bi.qp = ldg(P.qp + (idx));
  bi.log2cu = (uint8_t)log2cu;
  bi.slice = (uint16_t)sidx;

  // ---- motion ---------------------------------------------------------------------------------------------------
  if (!intra) {
    const int r0 = ldg(P.ref_idx[0] + (idx)), r1 = ldg(P.ref_idx[1] + (idx));
    int use0 = r0 >= 0, use1 = r1 >= 0;
    if (use0) { bi.mv[0][0] = ldg(P.mv[0] + (idx * 2)); bi.mv[0][1] = ldg(P.mv[0] + (idx * 2 + 1)); bi.ref[0] = ldg(&sl->ref_pic[0][r0]); }
    if (use1) { bi.mv[1][0] = ldg(P.mv[1] + (idx * 2)); bi.mv[1][1] = ldg(P.mv[1] + (idx * 2 + 1)); bi.ref[1] = ldg(&sl->ref_pic[1][r1]); }
    // xCheckIdenticalMotion (TComPrediction.cpp:497-512): B slice, both lists, same POC and same MV -> list 0 only
    if (ldg(&sl->slice_type) == HMGPU_B_SLICE && use0 && use1 && ldg(&sl->ref_poc[0][r0]) == ldg(&sl->ref_poc[1][r1]) &&
        bi.mv[0][0] == bi.mv[1][0] && bi.mv[0][1] == bi.mv[1][1])
      use1 = 0;
    bi.flags |= (use0 ? BF_MC_L0 : 0) | (use1 ? BF_MC_L1 : 0);
    atomicAdd(&lds_stat[1], 1u);
  } else {
    atomicAdd(&lds_stat[0], 1u);
  }

  // ---- deblocking edge flags (TComLoopFilter.cpp:269-409) ----------------------------------------------------------
  const int lf_across_slices = ldg(&sl->lf_across_slices);
  if (!ldg(&sl->deblocking_disable)) {
    const int tux = x4 & ~(tu_parts - 1), tuy = y4 & ~(tu_parts - 1);
    // vertical edge at the left border of this partition
    if ((px & 7) == 0) {
      bool filt, trans;
      if (rx == 0) {                     // CU border: m_stLFCUParam.bLeftEdge
        bool avail = px != 0;
        if (avail && x4 == 0) {          // crosses into the left CTU: getPULeft slice/tile restrictions
          const int n = ctu - 1;
          if (!lf_across_slices && P.slice_idx && ldg(P.slice_idx + (n)) != sidx) avail = false;
          if (!P.lf_across_tiles && P.tile_idx && ldg(P.tile_idx + (n)) != ldg(P.tile_idx + (ctu))) avail = false;
        }
        filt = trans = avail;
      } else {
        trans = x4 == tux;               // TU border inside the CU
        bool pu = false;                 // PU border inside the CU (xSetEdgefilterPU)
        switch (part_size) {
          case HMGPU_SIZE_Nx2N: case HMGPU_SIZE_NxN: pu = rx == (cu_parts >> 1); break;
          case HMGPU_SIZE_nLx2N: pu = rx == (cu_parts >> 2); break;
          case HMGPU_SIZE_nRx2N: pu = rx == cu_parts - (cu_parts >> 2); break;
          default: break;
        }
        filt = trans || pu;
      }
      bi.edge |= (filt ? BE_VER_FILTER : 0) | (trans ? BE_VER_TRANSFORM : 0);
    }
    if ((py & 7) == 0) {
      bool filt, trans;
      if (ry == 0) {
        bool avail = py != 0;
        if (avail && y4 == 0) {
          const int n = ctu - P.ctus_w;
          if (!lf_across_slices && P.slice_idx && ldg(P.slice_idx + (n)) != sidx) avail = false;
          if (!P.lf_across_tiles && P.tile_idx && ldg(P.tile_idx + (n)) != ldg(P.tile_idx + (ctu))) avail = false;
        }
        filt = trans = avail;
      } else {
        trans = y4 == tuy;
        bool pu = false;
        switch (part_size) {
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
  stg4(out, __builtin_bit_cast(u32x4, bi));

  // ---- transform units -------------------------------------------------------------------------------------------
  // cbf bit d of a partition = cbf of its ancestor TU node at transform depth d (TComDataCU.h:310); HM descends
  // only while every node on the way has its bit set (TComTrQuant.cpp:1558-1564)
  const unsigned chain = (1u << (tr + 1)) - 1;
  const int qp_cu = ldg(P.qp + ((size_t)ctu * parts + (z & ~(cu_parts * cu_parts - 1))));   // cu.getQP(0): first partition of the CU
  const int ctu_luma = 1 << (2 * P.log2ctu);
  if (log2tu > 5) return;                                   // not a legal HEVC TU size; nothing to transform
  if (intra) return;                                        // intra CUs are not reconstructed on the device yet: their
                                                            // residual must not be added to samples the caller supplied
  if (log2tu > 2) {
    if (x4 == (x4 & ~(tu_parts - 1)) && y4 == (y4 & ~(tu_parts - 1))) {
      if ((cbf_y & chain) == chain) {
        const int ts = P.tskip[0] ? ldg(P.tskip[0] + (idx)) : 0;
        stage_tu(P, st, lds_cnt, log2tu, gx, gy, 0, (ts ? 2 : 0), qp_cu, sl, (uint32_t)ctu * ctu_luma + 16u * z);
      }
      for (int comp = 1; comp < 3; comp++) {
        if ((ldg(P.cbf[comp] + (idx)) & chain) == chain) {
          const int ts = P.tskip[comp] ? ldg(P.tskip[comp] + (idx)) : 0;
          stage_tu(P, st, lds_cnt, log2tu - 1, gx, gy, comp, (ts ? 2 : 0), qp_cu, sl, (uint32_t)ctu * (ctu_luma >> 2) + 4u * z);
        }
      }
    }
  } else {
    // 4x4 luma TU: every partition is an origin
    if ((cbf_y & chain) == chain) {
      const int ts = P.tskip[0] ? ldg(P.tskip[0] + (idx)) : 0;
      stage_tu(P, st, lds_cnt, 2, gx, gy, 0, (intra ? 1 : 0) | (ts ? 2 : 0), qp_cu, sl, (uint32_t)ctu * ctu_luma + 16u * z);
    }
    // the one 4x4 chroma TU of the parent 8x8 node rides with the first child (z multiple of 4)
    if ((z & 3) == 0) {
      for (int comp = 1; comp < 3; comp++) {
        if ((ldg(P.cbf[comp] + (idx)) & chain) == chain) {
          const int ts = P.tskip[comp] ? ldg(P.tskip[comp] + (idx)) : 0;
          stage_tu(P, st, lds_cnt, 2, gx, gy, comp, (ts ? 2 : 0), qp_cu, sl, (uint32_t)ctu * (ctu_luma >> 2) + 4u * z);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_prep(const PicDev* __restrict__ pics, Batch b) {
  __shared__ uint32_t lds_cnt[4], lds_base[4], lds_stat[2];
  const PicDev& P = pics[b.pic[blockIdx.z]];
  if (threadIdx.x < 4) lds_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 2) lds_stat[threadIdx.x] = 0;
  __syncthreads();
  TuStage st; st.n = 0;
  const int gpart = blockIdx.x * 256 + threadIdx.x;
  if (gpart < b.num_ctus[blockIdx.z] * P.parts) prep_partition(P, b, gpart, st, lds_cnt, lds_stat);
  __syncthreads();
  const int shard = blockIdx.x & (kTuShards - 1);
  if (threadIdx.x < 4) {
    const uint32_t n = lds_cnt[threadIdx.x];
    lds_base[threadIdx.x] = n ? atomicAdd(&P.tu_count[threadIdx.x * kTuShards + shard], n) : 0u;
  } else if (threadIdx.x < 6) {
    const uint32_t n = lds_stat[threadIdx.x - 4];
    if (n) atomicAdd(&P.stats[(threadIdx.x - 4) * kTuShards + shard], (unsigned long long)n);
  }
  __syncthreads();
  for (int k = 0; k < st.n; k++) {
    const int c = st.cls[k];
    const uint32_t i = lds_base[c] + (uint32_t)st.loc[k];
    if (i < P.tu_cap[c]) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(P.tu[c] + (size_t)shard * P.tu_cap[c] + i);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&st.r[k]);
      stg(dst, src[0]); stg(dst + 1, src[1]); stg(dst + 2, src[2]);
    }
  }
}

// resets the TU list lengths of every picture of the batch (one launch instead of one memset per picture)
__global__ void k_zero_counts(const PicDev* __restrict__ pics, Batch b) {
  const PicDev& P = pics[b.pic[blockIdx.x]];
  if (threadIdx.x < 4 * kTuShards) stg(P.tu_count + threadIdx.x, 0u);
}

void launch_prep(const PicDev* pics, const Batch& b, int max_ctus, int parts, hipStream_t s) {
  hipLaunchKernelGGL(k_zero_counts, dim3((unsigned)b.n), dim3(64), 0, s, pics, b);
  dim3 grid((unsigned)(((size_t)max_ctus * parts + 255) / 256), 1, (unsigned)b.n);
  hipLaunchKernelGGL(k_prep, grid, dim3(256), 0, s, pics, b);
}

}  // namespace hmgpu
