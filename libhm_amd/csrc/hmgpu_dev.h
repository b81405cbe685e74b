// hmgpu_dev.h -- device-side data model shared by the HIP kernels and the host runtime of libhmgpu.so.
//
// Design: the host hands over HM's per-CTU TComDataCU arrays untouched (z-scan SoA).  A
// "prep" kernel turns them, one thread per 4x4 partition, into
//   * a picture-wide raster grid of 16-byte BlkInfo records (motion, QP, intra/cbf flags, deblocking edge
//     flags) that the motion-compensation, deblocking and SAO kernels index by sample position, and
//   * per-size lists of coded transform units (TuRec) appended with wave-aggregated atomics.
// No quadtree is ever walked serially; every kernel is a flat grid over partitions, tiles, TUs or edges.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hmgpu.h"

namespace hmgpu {

constexpr int kMaxBatch = 16;       // pictures per batched launch (blockIdx.z)
constexpr int kMaxPics = 64;        // device pictures per context
constexpr int kTuShards = 8;        // TU lists are sharded by (prep block index % 8) so that list appends do not pile on one word
// Chroma planes (round 4): Cb and Cr of a picture live in ONE plane, sample by sample -- Cb(x, y) at element y * pitch + 2 * x, Cr(x, y) one
// element on -- so that a window row of both components is one piece of one 128-byte line for motion compensation instead of a piece
// in each of two lines, and a row of reconstructed chroma leaves a wave as whole lines (measured on k_mc_chroma, timing only: 0.132 ->
// 0.107 ms per 16 pictures).  PicDev::rec[2] = rec[1] + 1, pitch[1] = pitch[2] = the pitch of the pair plane in int16 elements: sample
// (x, y) of component c > 0 is rec[c][y * pitch[c] + kCStep * x] everywhere.  The C ABI keeps HM's three planes (upload / download /
// packing / hashing take the components apart).
constexpr int kCStep = 2;

// ---- per-4x4-block record (raster grid over the CTU-padded picture) -------------------------------------------
struct __attribute__((aligned(16))) BlkInfo {
  int16_t mv[2][2];   // HM TComMv per list {hor, ver} in quarter luma samples, UNclipped (deblocking compares these); 0 if unused
  int8_t  ref[2];     // device picture handle per list (slice ref_pic[list][ref_idx]); -1 = list unused
  int8_t  qp;         // TComDataCU::getQP(partition)
  uint8_t flags;      // BF_*
  uint8_t edge;       // BE_*
  uint8_t log2cu;     // log2 CU size (3..6): clipMv needs the CU origin
  uint16_t slice;     // index into the picture's slice table
};
static_assert(sizeof(BlkInfo) == 16, "BlkInfo must be 16 bytes");

enum : uint8_t {
  BF_VALID = 1,    // partition decoded (part_size != NUMBER_OF_PART_SIZES) and inside the picture
  BF_INTRA = 2,
  BF_CBFY = 4,     // luma cbf at the partition's transform depth (Bs = 1 rule)
  BF_MC_L0 = 8,    // prediction uses list 0  (after the identical-motion collapse of xCheckIdenticalMotion)
  BF_MC_L1 = 16,   // prediction uses list 1
  BF_NOFILT = 32,  // lossless CU, or PCM CU with pcm_loop_filter_disabled: deblocking and SAO leave its samples alone
};
enum : uint8_t {
  BE_VER_FILTER = 1,     // left edge of this block is a deblocking edge (m_aapbEdgeFilter[EDGE_VER]) on the 8x8 grid
  BE_VER_TRANSFORM = 2,  // ... and it is a TU/CU edge (the m_aapucBS marker xGetBoundaryStrengthSingle reads)
  BE_HOR_FILTER = 4,
  BE_HOR_TRANSFORM = 8,
};

// ---- deblocking decisions of one 8x8 luma area (raster grid, grid_w/2 x grid_h/2), written by k_prep for the loop-filter kernels ----
// The four 4-sample edge units on the area's left border (v[k]: rows 4k .. 4k+3) and top border (h[k]: columns 4k .. 4k+3), 2 bytes
// each: what xGetBoundaryStrengthSingle, xEdgeFilterLuma and xEdgeFilterChroma take from the two blocks that face each other
// across the unit (TComLoopFilter.cpp:411-537, 587-600, 629-634) -- SURVEY.md a-12.  The slice constants of a unit are those of the
// CTU its Q side lies in (slices are whole CTUs).
//   bits 0-1  Bs; 0 = the unit is not filtered (no edge, deblocking disabled, unavailable neighbour or Bs 0)
//   bits 2-8  ((QP_P + QP_Q + 1) >> 1) + 32
//   bit 9/10  the P / Q side is exempt from the loop filters (lossless CU, PCM CU with pcm_loop_filter_disabled)
struct __attribute__((aligned(8))) EdgeRec { uint16_t v[2], h[2]; };
static_assert(sizeof(EdgeRec) == 8, "EdgeRec must be 8 bytes");
__host__ __device__ constexpr uint32_t edge_unit_pack(int bs, int qp, bool p_nf, bool q_nf) {
  return (uint32_t)bs | ((uint32_t)(qp + 32) << 2) | ((p_nf ? 1u : 0u) << 9) | ((q_nf ? 1u : 0u) << 10);
}

// ---- motion of one 8x8 luma tile (raster grid, grid_w/2 x grid_h/2), written by k_prep for the motion-compensation kernels --
// A tile is ACTIVE when its four 4x4 cells are inter-predicted with identical motion (k_mc.hip predicts those); slot 0 is
// the first list the tile predicts from, slot 1 is list 1 of a bi-predicted tile.  The vectors are CLIPPED (TComDataCU::clipMv)
// and split into integer luma samples and the quarter-sample fraction; unused fields are 0, so that two tiles continue each
// other vertically iff their first 12 bytes are equal.
struct __attribute__((aligned(16))) TileMv {
  int16_t ix0, iy0, ix1, iy1;   // integer parts of the clipped motion vectors of slot 0 / slot 1 (luma samples)
  uint8_t frac;                 // xf0 | yf0 << 2 | xf1 << 4 | yf1 << 6 (quarter samples)
  uint8_t ref0, ref1;           // device picture handles of slot 0 / slot 1
  uint8_t flags;                // TM_*
  uint8_t ridx;                 // reference indices of slot 0 | slot 1 << 4 (explicit weighted prediction)
  uint8_t rmask;                // TR_*: which parts of the tile carry a residual (written for every tile, active or not)
  uint16_t slice;               // index into the picture's slice table
};
static_assert(sizeof(TileMv) == 16, "TileMv must be 16 bytes");
enum : uint8_t { TM_ACTIVE = 1, TM_BI = 2, TM_FIRST_L1 = 4 };
__host__ __device__ constexpr int resid_slot(int row) { return ((row & 1) << 2) | ((row & 7) >> 1); }   // 16-byte slot of row `row` in its tile
// residual mask of an 8x8 luma tile: bits 0-3 its four 4x4 luma quadrants (raster order), bit 4 / 5 the 4x4 Cb / Cr block under it
enum : uint8_t { TR_LUMA = 15, TR_CB = 16, TR_CR = 32 };

// ---- coded transform unit ---------------------------------------------------------------------------------------
struct TuRec {
  uint16_t x4, y4;        // origin in 4-luma-sample units (picture coordinates)
  uint8_t  comp_flags;    // bits 0-1 component, bit 2 DST (4x4 intra luma), bit 3 transform skip, bit 4 cu_transquant_bypass, bits 5-6 RDPCM
                          // (1 horizontal, 2 vertical: explicit for inter blocks, implied by the prediction mode for intra ones), bit 7 intra CU
  int8_t   per, rem;      // QpParam per / rem
  uint8_t  xflags;        // bit 0: the 4x4 block is read back to front (transform_skip_rotation, intra)
  uint32_t coef_off;      // element offset into the component's coefficient array
};
static_assert(sizeof(TuRec) == 12, "TuRec must be 12 bytes");

// ---- slice constants on the device --------------------------------------------------------------------------------
struct SliceDev {
  int32_t slice_type;
  int32_t cb_qp_offset, cr_qp_offset;
  int32_t pps_cb_qp_offset, pps_cr_qp_offset;
  int32_t deblocking_disable, beta_offset_div2, tc_offset_div2, lf_across_slices;
  int32_t ref_poc[2][HMGPU_MAX_REF];
  int8_t  ref_pic[2][HMGPU_MAX_REF];
  int32_t constrained_intra_pred;
  int32_t weighted_pred;                // explicit weighted prediction active (TComSlice::applyWP)
  int32_t wp_log2_denom[2];
  int16_t wp_weight[2][HMGPU_MAX_REF][3];
  int16_t wp_offset[2][HMGPU_MAX_REF][3];
};

struct SaoDev {                 // reconstructed SAO parameters of one CTU component, 12 bytes
  int8_t  type;                 // -1 off, else HMGPU_SAO_EO_0..BO
  uint8_t band;                 // BO: first band
  uint16_t avail;               // 3x3 grid around the CTU, bit 3 * v + h (v: 0 above, 1 same row, 2 below; h: 0 left, 1 same
                                // column, 2 right): that CTU may be read by SAO (bit 4, the CTU itself, is always set)
  int8_t  off[8];               // EO: [0..4] by edge class (class 2 == 0); BO: [0..3] = offsets of bands band+0..3 (mod 32)
};
static_assert(sizeof(SaoDev) == 12, "SaoDev layout");

struct PlaneSet { int16_t* p[3]; };

// ---- everything the kernels need to know about one picture; one slot per device picture, resident in HBM ----------
struct PicDev {
  int32_t width, height;           // luma samples
  int32_t bd[3];                   // bit depth per component
  int32_t log2ctu, ctus_w, ctus_h, num_ctus, parts, pw;   // pw = partitions per CTU row
  int32_t pitch[3];                // int16 elements per row; chroma: of the plane that holds both components (kCStep)
  int32_t mx[3], my[3];            // margins (samples / rows) around every plane, border-extended like TComPicYuv::extendPicBorder
  int32_t grid_w, grid_h;          // BlkInfo grid (CTU padded)
  int32_t lf_across_tiles;
  int32_t sao_applied;             // final planes are sao[] (else rec[])
  int16_t* rec[3];                 // reconstruction / deblocked in place ([2] = [1] + 1: Cb and Cr alternate in one plane)
  int16_t* sao[3];                 // SAO output
  // raw HM arrays (device copies, whole picture)
  const uint8_t* depth; const int8_t* part_size; const int8_t* pred_mode; const int8_t* qp; const uint8_t* tr_idx;
  const uint8_t* cbf[3]; const uint8_t* tskip[3];
  const int16_t* mv[2]; const int8_t* ref_idx[2];
  const uint8_t* intra_dir[2];     // m_puhIntraDir[luma, chroma]
  const uint8_t* bypass; const uint8_t* ipcm;   // m_CUTransquantBypass, m_pbIPCMFlag
  const int16_t* pcm[3];           // PCM sample buffers (allocated on first use), layout of coef[]
  int32_t pcm_shift[3];            // bit depth - PCM bit depth per component
  int32_t pcm_lf_disable;          // SPS pcm_loop_filter_disabled_flag (with PCM enabled)
  int32_t any_nofilt;              // some partition of the picture carries BF_NOFILT (host-side scan): SAO looks at the flags
  const uint16_t* slice_idx; const uint16_t* tile_idx;
  const int16_t* coef[3];
  // residual of the inter TUs, written by k_itx and added by the motion-compensation kernels: per component 8x8-sample tiles of 128
  // bytes (8 rows x 16 bytes), tile (tx, ty) at index ty * (padded component width / 8) + tx; inside a tile the even rows come first
  // (resid_slot): a motion-compensation lane owns rows 2q, 2q+1, so each of its two loads reads 64 consecutive bytes per tile
  int16_t* resid[3];
  const uint32_t* coef_start[3];   // compact levels: element offset of every CTU's first coded TU (+ total), else null (HM's dense layout)
  uint32_t* quad_off[3];           // compact levels: offset of the first TU that starts in every 8x8 luma area (z-order), written by k_prep
  const SliceDev* slices;
  // derived
  BlkInfo* blk;                    // written only for calls that run kernels which read it (mixed-motion tiles, exempt CUs): launch_prep
  EdgeRec* edges;                  // [grid_h / 2][grid_w / 2]
  TileMv* tmv;                     // [grid_h / 2][grid_w / 2]
  TuRec* tu[4];                    // by log2 size - 2: kTuShards shards of tu_cap[] records each
  uint32_t* tu_count;              // [4][kTuShards]
  uint32_t tu_cap[4];              // capacity of ONE shard
  SaoDev* saoprm;                  // [num_ctus][3]
  unsigned long long* stats;       // [2][kTuShards]: intra / inter partitions seen by the prep kernel
  // scaling lists: one byte per position, [size 4x4..32x32][list = 3 * inter + component][1024], or null (flat)
  const uint8_t* sl_m;
  // intra reconstruction (k_intra.hip)
  int32_t has_intra_dir;           // the caller supplied intra prediction modes (else intra CUs are left untouched)
  int32_t strong_intra_smoothing;  // SPS flag
  int32_t range_ext;               // HMGPU_REXT_* (sps_range_extension tools of the residual path)
  int32_t mono;                    // chroma_format_idc 0: no chroma blocks are coded, the chroma planes are never read back
  int32_t fmt, csx, csy;           // chroma_format_idc (0 kept as 1) and the chroma subsampling it implies (getComponentScaleX / Y, TComChromaFormat.h:59-62);
                                   // 4:2:2 / 4:4:4 pictures take their chroma through the format-generic kernels of k_cfmt.hip
  const int8_t* ccp[2];            // cross-component prediction weights per partition (Cb, Cr), or null
  uint8_t* ctu_intra;              // [num_ctus] number of the CTU's 8x8 areas that belong to intra CUs, 0 = none (written by k_prep)
  uint32_t* intra_done;            // [3][num_ctus]: the CTU's intra CUs of that component are reconstructed
  uint32_t* fault;                 // set by a kernel that gave up waiting (k_intra's bounded spin): checked by the host at hmgpu_sync
  int32_t debug_skip_ctu;          // test hook (hmgpu_debug_stall_intra): the intra workgroups of this CTU leave without reconstructing or publishing (-1: none)
};

// batched launch descriptor, passed by value
struct Batch {
  int32_t n;
  int32_t pic[kMaxBatch];          // index into the PicDev table
  int32_t first_ctu[kMaxBatch];
  int32_t num_ctus[kMaxBatch];
};

// everything the motion-compensation kernels need about a batch, in kernel-argument memory: nothing stands between a workgroup's
// start and the load of its tile records (k_mc.hip).  Filled by the host from its mirrors of the PicDev descriptors.
struct McArgs {
  int32_t n, mode, log2n, per;                 // pictures of the batch; grid mapping (launch_mc)
  int32_t width, height, log2ctu, ctus_w;      // geometry (the same for every picture of a context)
  int32_t tw, pitch, bd, npics;                // TileMv grid width; pitch / bit depth of the component; entries of finals[]
  int32_t first_ctu[kMaxBatch], num_ctus[kMaxBatch];
  const TileMv* tmv[kMaxBatch];
  int16_t* dst[kMaxBatch];                     // luma: the picture's reconstruction plane; chroma: Cb
  int16_t* dst2[kMaxBatch];                    // chroma: Cr
  const int16_t* resid[kMaxBatch];             // residual tiles of the component (luma / Cb), PicDev::resid
  const int16_t* resid2[kMaxBatch];            // chroma: Cr
  int32_t rtw, pad_;                           // residual tiles per tile row of the component
  const SliceDev* slices[kMaxBatch];           // slice tables (explicit weighted prediction)
  // final planes: every picture of a context lives in one slab, picture i at slab + i * pic_stride; its final luma plane (sample
  // (0,0)) is at + origin_off, or + sao_off + origin_off when bit i of the SAO mask is set (the picture went through SAO)
  const char* slab;
  uint64_t pic_stride, slab_bytes;
  uint32_t sao_off, origin_off, cr_off;        // origin_off: luma launch: Y plane; chroma launch: Cb plane; cr_off: Cb -> Cr
  uint32_t sao_mask_lo, sao_mask_hi;
};

// the same for the residual kernel (k_itx.hip)
struct ItxArgs {
  int32_t n; uint32_t class_mask;              // pictures of the batch; size classes to run (bit = log2 size - 2)
  int32_t blocks[4];                           // workgroups per shard and size class (launch_itx)
  int32_t rtw[3], bd[3];                       // residual tiles per tile row; bit depths
  int32_t csx, csy;                            // chroma subsampling (a TU record carries its LUMA position)
  uint32_t tu_cap[4];                          // capacity of one shard's list
  const TuRec* tu[kMaxBatch][4];
  const uint32_t* tu_count[kMaxBatch];         // [4][kTuShards]
  const int16_t* coef[kMaxBatch][3];
  int16_t* resid[kMaxBatch][3];
  const uint8_t* sl_m[kMaxBatch];
};

// Pointers stored inside PicDev / PlaneSet reach the kernels through memory, so the compiler only knows them as generic
// ("flat") pointers: flat loads cannot be counted separately from LDS traffic and force a full s_waitcnt after every
// batch.  Every hot pointer is therefore re-typed as a global-address-space pointer before use.
#define HMGPU_AS1 __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
template <typename T> __device__ inline T ldg(const T* p) { return *(const T HMGPU_AS1*)p; }          // scalar types
template <typename T> __device__ inline void stg(T* p, T v) { *(T HMGPU_AS1*)p = v; }
__device__ inline u32x4 ldg4(const void* p) { return *(const u32x4 HMGPU_AS1*)p; }
__device__ inline u32x4 ldg4_a8(const void* p) { return *(const u32x4_a8 HMGPU_AS1*)p; }
__device__ inline u32x2 ldg2(const void* p) { return *(const u32x2 HMGPU_AS1*)p; }
__device__ inline void stg4(void* p, u32x4 v) { *(u32x4 HMGPU_AS1*)p = v; }
__device__ inline void stg4_a8(void* p, u32x4 v) { *(u32x4_a8 HMGPU_AS1*)p = v; }
__device__ inline void stg2(void* p, u32x2 v) { *(u32x2 HMGPU_AS1*)p = v; }

// ---- small device helpers ------------------------------------------------------------------------------------------
__host__ __device__ inline int zscan_x(int z) {   // HM g_auiZscanToRaster column: even bits of z
  int x = z & 0x5555; x = (x | (x >> 1)) & 0x3333; x = (x | (x >> 2)) & 0x0f0f; x = (x | (x >> 4)) & 0x00ff; return x;
}
__host__ __device__ inline int zscan_y(int z) { return zscan_x(z >> 1); }
__device__ inline int clip3(int lo, int hi, int v) { return min(hi, max(lo, v)); }
__device__ inline BlkInfo ld_blk(const BlkInfo* p) { return __builtin_bit_cast(BlkInfo, ldg4(p)); }

// XCD-aware work distribution (speed only, never correctness): workgroups are dealt round-robin to the 8 XCDs, each
// with its own L2.  For a batch of n pictures with nb workgroups each, picture p is served by the XCDs x with
// x % n == p and each of those XCDs walks one contiguous band of the picture in raster order, so that neighbouring
// tiles (which share reference-picture rows) hit in the same L2.  Returns false for padding workgroups.
__device__ inline bool xcd_remap(int bid, int n, int nb, int& slot, int& lb) {
  if (n < 0) { n = -n; slot = bid % n; lb = bid / n; return lb < nb; }     // tuning knob: plain interleave
  if (n <= 8 && (8 % n) == 0) {
    const int m = 8 / n;                       // XCDs per picture
    const int per = (nb + m - 1) / m;          // workgroups per band
    const int x = bid & 7, t = bid >> 3;
    slot = x % n;
    const int r = x / n;
    lb = r * per + t;
    return t < per && lb < nb;
  }
  slot = bid % n; lb = bid / n;
  return lb < nb;
}
__host__ inline int xcd_grid(int n, int nb) {
  if (n <= 8 && (8 % n) == 0) { const int m = 8 / n; return 8 * ((nb + m - 1) / m); }
  return n * nb;
}

// ---- launchers (one per kernel family; defined in the .hip files) ---------------------------------------------------
void launch_prep(const PicDev* pics, const Batch& b, int max_ctus, int parts, bool intra, bool write_blk, int fmt, hipStream_t s);
// npics = entries of the finals table (device pictures of the context, <= kMaxPics)
// bi: the batch holds B slices (the variants that run the H and V passes once per list)
void launch_mc_luma(McArgs& a, int max_ctus, bool wp, bool bi, hipStream_t s);
void launch_mc_chroma(McArgs& a, int max_ctus, bool wp, bool bi, hipStream_t s);
// the 4x4 cells of tiles with mixed motion (8x4 / 4x8 PUs, AMP in 16x16 CUs): only for calls that contain such PUs
void launch_mc_luma_cells(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, bool wp, hipStream_t s);
void launch_mc_chroma_cells(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, bool wp, hipStream_t s);
void launch_itx(ItxArgs& a, uint32_t blocks_per_shard, hipStream_t s);
void launch_filter_fused(const PicDev* pics, const Batch& b, int width, int height, bool nofilt, hipStream_t s);
// step: distance of two samples of the plane in memory (1 luma, kCStep chroma)
void launch_pack(const int16_t* src, int pitch, int step, int x0, int y0, int w, int h, int bytes, uint8_t* dst, int dst_stride, hipStream_t s);
void launch_unpack(const int16_t* src, int w, int h, int16_t* dst, int pitch, int step, hipStream_t s);
void launch_checksum(const int16_t* src, int pitch, int step, int w, int h, int bd, uint32_t* out, hipStream_t s);
// MD5 chains, one per lane (k_out.hip): message, length, where the four state words a, b, c, d go
struct Md5Batch { int32_t n, pad_; const uint8_t* msg[128]; unsigned long long bytes[128]; uint32_t* out[128]; };
void launch_md5(const Md5Batch& job, hipStream_t s);
void launch_crc(const int16_t* src, int pitch, int step, int w, int h, int bd, uint32_t* rows, uint32_t* out, hipStream_t s);
void launch_intra(const PicDev* pics, const Batch& b, const int32_t* order, int num_ctus, bool lean, hipStream_t s);   // lean: no I slices in the call
// chroma of 4:2:2 / 4:4:4 pictures (k_cfmt.hip): cross-component prediction on the residual tiles, motion compensation of every inter
// cell, chroma deblocking on the format's own grid; fmt = chroma_format_idc
void launch_ccp(const PicDev* pics, const Batch& b, int max_ctus, hipStream_t s);
void launch_mc_chroma_fmt(const PicDev* pics, const PlaneSet* finals, const Batch& b, int max_ctus, int log2ctu, int fmt, bool wp, hipStream_t s);
void launch_deblock_chroma_fmt(const PicDev* pics, const Batch& b, int dir, int width, int height, hipStream_t s);
void launch_intra_chroma_422(const PicDev* pics, const Batch& b, hipStream_t s);
void launch_deblock(const PicDev* pics, const Batch& b, int dir, int width, int height, hipStream_t s);
void launch_sao(const PicDev* pics, const Batch& b, int width, int height, int csx, int csy, hipStream_t s);
void launch_extend(const PicDev* pics, const Batch& b, int width, int height, int mx, int my, int csx, int csy, hipStream_t s);
// kernel-level seams for tests
void launch_itx_flat(int log2size, int bit_depth, int n, const int16_t* levels, const int8_t* per, const int8_t* rem,
                     const uint8_t* flags, int16_t* resid, hipStream_t s);
void launch_mc_flat(int is_chroma, int bit_depth, const int16_t* ref, int ref_stride, int ref_w, int ref_h, int n,
                    const int32_t* blocks, const int32_t* out_off, int bi, int16_t* dst, hipStream_t s);

}  // namespace hmgpu
