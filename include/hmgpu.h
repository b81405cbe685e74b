/* hmgpu.h -- C ABI of libhmgpu.so: MI355X-native pixel reconstruction for the HM (HEVC) decoder.
 *
 * This is the drop-in boundary for the one hot path of ChristianFeldmann/libHM (HM 16.0) that is
 * accelerated: everything HM does between the end of CABAC parsing of a slice and the finished,
 * loop-filtered picture.  HM has no plugin seam there, so the seam sits at the two
 * calls HM itself isolates and times:
 *
 *     TDecGop::decompressSlice()  TLibDecoder/TDecGop.cpp:105   -> hmgpu_decompress_slice()
 *     TDecGop::filterPicture()    TLibDecoder/TDecGop.cpp:157   -> hmgpu_filter_picture()
 *
 * Parsing stays with the (host) caller.  What crosses the boundary is exactly what HM's parser leaves
 * behind in TComPicSym: per-CTU TComDataCU arrays (4x4-partition granularity, z-scan order,
 * TComDataCU.h:86-157), coefficient levels (TComDataCU.cpp:165-173), SAOBlkParam per CTU
 * (TypeDef.h:754-779) and a handful of slice/PPS/SPS constants that HM keeps in globals.  The
 * reference-side shim is a field-by-field memcpy.  All traversal (CU/TU/PU
 * quadtrees, boundary strengths, SAO merge resolution ...) happens behind this interface, on the GPU.
 *
 * Conventions: plain C, no HIP/torch types; every function returns an hmgpu_status (never aborts,
 * unlike HM's assert/exit: TComTrQuant.cpp:925); one context per host thread and per GPU; calls on one
 * context are serialised by the caller; work is enqueued on the context's HIP stream and completes at
 * hmgpu_sync()/hmgpu_picture_download().  Samples are HM's Pel = int16, levels int16 (HM's TCoeff
 * int32 narrowed by the shim; exact for bit depth <= 10 because xDeQuant clips its input to 16 bits
 * first: TComTrQuant.cpp:1284-1287).
 */
#ifndef HMGPU_H
#define HMGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMGPU_VERSION 1

typedef enum hmgpu_status {
  HMGPU_OK = 0,
  HMGPU_EINVAL = 1,       /* bad argument / inconsistent geometry (HM: assert) */
  HMGPU_EDEVICE = 2,      /* HIP error; code via hmgpu_last_device_error() */
  HMGPU_EUNSUPPORTED = 3, /* coding tool outside the supported envelope; nothing was enqueued */
  HMGPU_ENOMEM = 4
} hmgpu_status;

#define HMGPU_MAX_REF 16         /* MAX_NUM_REF in HM (CommonDef.h) */
#define HMGPU_MAX_SLICES 600     /* slices per picture kept on the device */
#define HMGPU_NO_PIC (-1)

/* HM enums that appear in the metadata (TypeDef.h:374-443) */
enum { HMGPU_B_SLICE = 0, HMGPU_P_SLICE = 1, HMGPU_I_SLICE = 2 };
enum { HMGPU_MODE_INTER = 0, HMGPU_MODE_INTRA = 1 };
enum { HMGPU_SIZE_2Nx2N = 0, HMGPU_SIZE_2NxN = 1, HMGPU_SIZE_Nx2N = 2, HMGPU_SIZE_NxN = 3, HMGPU_SIZE_2NxnU = 4,
       HMGPU_SIZE_2NxnD = 5, HMGPU_SIZE_nLx2N = 6, HMGPU_SIZE_nRx2N = 7, HMGPU_SIZE_NONE = 8 /* NUMBER_OF_PART_SIZES: not decoded */ };
enum { HMGPU_SAO_OFF = 0, HMGPU_SAO_NEW = 1, HMGPU_SAO_MERGE = 2 };                   /* SAOMode, TypeDef.h:604 */
enum { HMGPU_SAO_EO_0 = 0, HMGPU_SAO_EO_90 = 1, HMGPU_SAO_EO_135 = 2, HMGPU_SAO_EO_45 = 3, HMGPU_SAO_BO = 4 }; /* :620 */
enum { HMGPU_SAO_MERGE_LEFT = 0, HMGPU_SAO_MERGE_ABOVE = 1 };                         /* :612 */

typedef struct hmgpu_ctx hmgpu_ctx;
typedef int32_t hmgpu_pic;       /* handle of a device-resident picture (TComPic/TComPicYuv counterpart) */

/* What TDecTop::xActivateParameterSets turns into globals (TDecTop.cpp:283-348: g_bitDepth, g_uiMaxCUWidth,
 * g_uiMaxCUDepth ...).  Fixed for the life of a context. */
typedef struct hmgpu_seq_params {
  int32_t width, height;          /* SPS pic_{width,height}_in_luma_samples (multiple of the 8x8 minimum CU) */
  int32_t bit_depth_luma;         /* g_bitDepth[CHANNEL_TYPE_LUMA]   (8..12; without extended_precision_processing) */
  int32_t bit_depth_chroma;       /* g_bitDepth[CHANNEL_TYPE_CHROMA] (8..12) */
  int32_t chroma_format;          /* chroma_format_idc: 1 (4:2:0), 2 (4:2:2), 3 (4:4:4), or 0 (4:0:0, monochrome: the chroma arrays hold no coded
                                     blocks, the chroma planes are allocated like those of 4:2:0 and left alone; picture hashes: the first
                                     digest).  Chroma planes, level arrays and PCM buffers have (width >> sx) x (height >> sy) samples per
                                     luma area, sx = 1 unless 4:4:4, sy = 1 only for 4:2:0 (getComponentScaleX/Y, TComChromaFormat.h:59-62) */
  int32_t log2_ctu_size;          /* log2 g_uiMaxCUWidth: 4, 5 or 6.  partitions are 4x4 => (1<<(2*log2_ctu_size-4)) per CTU */
  int32_t max_pictures;           /* device pictures to pre-allocate (DPB size + pictures in flight) */
  int32_t pcm_loop_filter_disable;/* SPS pcm_loop_filter_disabled_flag && pcm_enabled_flag */
  int32_t strong_intra_smoothing; /* SPS strong_intra_smoothing_enabled_flag (TComPattern.cpp:201-216) */
  int32_t pcm_bit_depth_luma, pcm_bit_depth_chroma;   /* SPS pcm_sample_bit_depth_*: PCM samples are shifted up to the coding bit depth (TDecCu.cpp:770-789) */
  int32_t range_ext_flags;        /* HMGPU_REXT_*: the sps_range_extension() tools that change reconstruction (0 for version-1 streams) */
  int32_t reserved[4];
} hmgpu_seq_params;

/* hmgpu_seq_params.range_ext_flags: what TComSPS keeps of sps_range_extension() for the residual path */
enum {
  HMGPU_REXT_ROTATION = 1,        /* getUseResidualRotation(): 4x4 intra transform-skip / bypass blocks are read back to front
                                     (TComTU::isNonTransformedResidualRotated, TComTU.cpp:227-233) */
  HMGPU_REXT_IMPLICIT_RDPCM = 2,  /* getUseResidualDPCM(RDPCM_SIGNAL_IMPLICIT): intra transform-skip / bypass blocks predicted
                                     horizontally (10) or vertically (26) accumulate their residual along that direction
                                     (invRdpcmNxN, TComTrQuant.cpp:1737-1792); bypass CUs also lose the intra edge filters
                                     (TComPrediction.cpp:476) */
  HMGPU_REXT_EXPLICIT_RDPCM = 4,  /* getUseResidualDPCM(RDPCM_SIGNAL_EXPLICIT): inter blocks carry their mode in transform_skip[] */
  HMGPU_REXT_INTRA_SMOOTHING_DISABLED = 8   /* getDisableIntraReferenceSmoothing(): intra reference samples are never filtered
                                               (TComPrediction::filteringIntraReferenceSamples, called from TDecCu.cpp:532) */
};

/* Scaling lists as TDecTop activates them for a slice (TDecTop.cpp:651-668: PPS lists, else SPS lists, else the defaults):
 * TComScalingList's own storage.  De-quantisation with them: TComTrQuant.cpp:1238-1275, tables :2992-3012, 3092-3106. */
typedef struct hmgpu_scaling_lists {
  int32_t coef[4][6][64];   /* getScalingListAddress(sizeId 4x4..32x32, listId = 3 * inter + component): raster order; 16 values for
                               4x4, 8x8 values otherwise (16x16 / 32x32 replicate every value over ratio x ratio positions) */
  int32_t dc[4][6];         /* getScalingListDC: replaces position 0 for 16x16 and 32x32 */
} hmgpu_scaling_lists;

/* Per-slice constants the hot path reads through pcCU->getSlice() */
typedef struct hmgpu_slice_params {
  int32_t slice_type;                   /* HMGPU_{B,P,I}_SLICE */
  int32_t cb_qp_offset, cr_qp_offset;   /* pps_cb/cr_qp_offset + slice_cb/cr_qp_offset: dequant QpParam (TComTrQuant.cpp:107-112) */
  int32_t pps_cb_qp_offset, pps_cr_qp_offset; /* PPS part only: chroma deblocking (TComLoopFilter.cpp:759) */
  int32_t deblocking_disable;           /* getDeblockingFilterDisable() */
  int32_t beta_offset_div2, tc_offset_div2;
  int32_t lf_across_slices;             /* getLFCrossSliceBoundaryFlag() */
  int32_t weighted_pred;                /* TComSlice::applyWP(): explicit weighted prediction is active for this slice (P slice with
                                           weighted_pred_flag, B slice with weighted_bipred_flag); the tables below are then read */
  int32_t lf_across_tiles;              /* PPS loop_filter_across_tiles_enabled_flag (TDecGop.cpp:165) */
  int32_t num_ref_idx[2];
  hmgpu_pic ref_pic[2][HMGPU_MAX_REF];  /* getRefPic(list, idx) as device picture handles */
  int32_t ref_poc[2][HMGPU_MAX_REF];    /* getRefPOC(list, idx) (identical-motion test, TComPrediction.cpp:497-512) */
  int32_t constrained_intra_pred;       /* PPS constrained_intra_pred_flag (TComPattern.cpp:558-572) */
  int32_t reserved[4];
  /* explicit weighted prediction (TComWeightPrediction.cpp:44-57, 211-271), after TComSlice::initWpScaling: per list,
   * reference index and component the weight (iWeight; 1 << log2 denominator where the header carried none) and the offset
   * already scaled to the bit depth (iOffset << (bitDepth - 8) unless high_precision_offsets) */
  int32_t wp_log2_denom[2];             /* luma, chroma */
  int16_t wp_weight[2][HMGPU_MAX_REF][3];
  int16_t wp_offset[2][HMGPU_MAX_REF][3];
  const hmgpu_scaling_lists* scaling_lists;   /* SPS scaling_list_enabled_flag: the lists in force, else NULL (flat, m = 16).  All slices of a
                                               * picture name one PPS (7.4.7.1), i.e. the same lists: the device keeps ONE table per picture,
                                               * written by every slice call */
} hmgpu_slice_params;

/* The picture-persistent TComDataCU arrays of TComPicSym (TComPicSym.cpp:93-114).  Every array covers the WHOLE
 * picture: [num_ctus][parts_per_ctu] in CTU raster order and HM z-scan order inside the CTU; a call only reads the
 * CTUs it is asked to process.  Optional arrays may be NULL (treated as all zero). */
typedef struct hmgpu_ctu_meta {
  const uint8_t* depth;              /* m_puhDepth */
  const int8_t*  part_size;          /* m_pePartSize (HMGPU_SIZE_*; HMGPU_SIZE_NONE = CTU part never decoded) */
  const int8_t*  pred_mode;          /* m_pePredMode */
  const int8_t*  qp;                 /* m_phQP */
  const uint8_t* tr_idx;             /* m_puhTrIdx */
  const uint8_t* cbf[3];             /* m_puhCbf[Y,Cb,Cr]: bit d = cbf at transform depth d (TComDataCU.h:310) */
  const uint8_t* transform_skip[3];  /* bit 0: m_puhTransformSkip[Y,Cb,Cr]; bits 1-2: m_explicitRdpcmMode[Y,Cb,Cr] of inter
                                        transform-skip / bypass blocks (RDPCM_OFF 0, RDPCM_HOR 1, RDPCM_VER 2: TypeDef.h)   (optional) */
  const int16_t* mv[2];              /* m_acCUMvField[list].m_pcMv as {hor,ver} pairs: [num_ctus][parts][2] */
  const int8_t*  ref_idx[2];         /* m_acCUMvField[list].m_piRefIdx (-1 = list unused) */
  const uint8_t* intra_dir[2];       /* m_puhIntraDir[luma,chroma]                          (optional; intra path) */
  const uint8_t* transquant_bypass;  /* m_CUTransquantBypass: lossless CUs (residual = levels, exempt from the loop filters)  (optional) */
  const uint8_t* ipcm;               /* m_pbIPCMFlag: PCM CUs (samples from coeffs->pcm_sample)                               (optional) */
  const uint16_t* slice_idx;         /* [num_ctus] index into the picture's slice table     (optional: all 0) */
  const uint16_t* tile_idx;          /* [num_ctus] TComPicSym::getTileIdxMap                (optional: all 0) */
  const int8_t*  ccp_alpha[2];       /* m_crossComponentPredictionAlpha[Cb, Cr] (4:4:4 with cross_component_prediction_enabled_flag: the chroma
                                        residual of a transform unit takes (alpha * luma residual) >> 3 on top, TComTrQuant.cpp:3294-3335)  (optional) */
} hmgpu_ctu_meta;

/* Coefficient levels, HM layout (m_pcTrCoeff: TU blocks contiguous in z-order, raster inside a TU, offset of a TU =
 * 16 * z-index of its first partition for luma, 4 * ... for chroma; TComTU.cpp:64-76,186): whole-picture arrays
 * y: [num_ctus][ctu*ctu], cb/cr: [num_ctus][ctu*ctu/4] (4:2:2: /2 -- a chroma block of a transform unit is two squares, the upper one
 * first, TComTU::VERTICAL_SPLIT, TComTrQuant.cpp:1436-1462; 4:4:4: /1).  In 4:2:2 cbf[1..2] carry, one transform depth below the unit's
 * own bit, the flags of the two squares over the upper / lower half of its partitions (TDecSbac.cpp:1058-1095). */
typedef struct hmgpu_coeffs {
  const int16_t* level[3];
  const int16_t* pcm_sample[3];   /* TComDataCU::getPCMSample (m_pcIPCMSample*): the transmitted samples of PCM CUs, same layout as
                                     the levels; needed only if meta->ipcm marks PCM CUs (may be NULL otherwise) */
  /* Compact levels (optional; all three NULL = HM's dense layout above).  level[c] then holds ONLY the coded transform units of
   * component c -- the blocks HM's layout would hold, in the same order (CTU by CTU, z-order of the TU origins inside a CTU, raster
   * inside a TU), with the uncoded ones left out; a TU is coded iff its cbf bits are set down to its transform depth
   * (TComTrQuant.cpp:1558-1564).  ctu_level_start[c][a] = element offset of CTU a's first coded TU, [num_ctus] = total: only that
   * many elements cross the bus.  Whole-picture calls only.  hmgpu_pack_levels() converts HM's arrays. */
  const uint32_t* ctu_level_start[3];
} hmgpu_coeffs;

/* SAOBlkParam as parsed (TDecSbac::parseSAOBlkParam), before reconstructBlkSAOParams: [num_ctus][3] */
typedef struct hmgpu_sao_param {
  int32_t mode_idc;       /* HMGPU_SAO_OFF / NEW / MERGE */
  int32_t type_idc;       /* NEW: HMGPU_SAO_EO_* / BO; MERGE: HMGPU_SAO_MERGE_LEFT / ABOVE */
  int32_t type_aux_info;  /* BO: first band */
  int32_t offset[32];     /* coded offsets (EO: classes 0..4; BO: by band) */
} hmgpu_sao_param;

typedef struct hmgpu_pic_params {
  int32_t lf_across_tiles;       /* PPS loop_filter_across_tiles_enabled_flag */
  int32_t sao_enabled;           /* SPS sample_adaptive_offset_enabled_flag (TDecGop.cpp:169) */
  int32_t sao_offset_shift_luma, sao_offset_shift_chroma;  /* log2_sao_offset_scale_* (0 for Main/Main10) */
  int32_t reserved[8];
} hmgpu_pic_params;

/* ------------------------------------------------------------------------------------------------ context */
hmgpu_status hmgpu_create(const hmgpu_seq_params* seq, int device_ordinal, hmgpu_ctx** out);
void         hmgpu_destroy(hmgpu_ctx* ctx);
int32_t      hmgpu_last_device_error(const hmgpu_ctx* ctx);        /* hipError_t of the last HMGPU_EDEVICE; -2: an intra wavefront gave up
                                                                       waiting for a neighbouring CTU (reported by hmgpu_sync / hmgpu_picture_download) */
/* Test hook for the error path above: from the next hmgpu_decompress_* call on `pic`, the intra wavefront leaves CTU `ctu` out (no samples,
 * no progress published), so the CTUs that predict from it run into their bounded wait and the call sequence ends in HMGPU_EDEVICE with
 * hmgpu_last_device_error() == -2 instead of hanging.  ctu = -1 switches it off.  Not for production use. */
hmgpu_status hmgpu_debug_stall_intra(hmgpu_ctx* ctx, hmgpu_pic pic, int32_t ctu);
const char*  hmgpu_status_string(hmgpu_status s);
hmgpu_status hmgpu_sync(hmgpu_ctx* ctx);                           /* wait for everything enqueued so far */

/* Page-locked host memory for the caller's arrays (metadata, levels, SAO parameters, download targets): staging from it is a true
 * asynchronous DMA at full link speed, from ordinary memory the runtime copies through a bounce buffer while the caller waits.
 * Returns NULL when no device memory manager is available (no GPU): use malloc then. */
void* hmgpu_host_alloc(size_t bytes);
void  hmgpu_host_free(void* p);

/* geometry helpers (TComPicSym::create, TComPicSym.cpp:73-92) */
int32_t hmgpu_num_ctus(const hmgpu_seq_params* seq);
int32_t hmgpu_parts_per_ctu(const hmgpu_seq_params* seq);

/* ------------------------------------------------------------------------------------------------ pictures
 * Counterpart of TDecTop::xGetNewPicBuffer (TDecTop.cpp:134): pictures live in HBM for as long as they are
 * referenced; planes never leave the device unless downloaded.  Host plane layout at upload/download is HM's
 * TComPicYuv convention reduced to the visible area: pointer to sample (0,0) + stride in samples. */
hmgpu_status hmgpu_picture_acquire(hmgpu_ctx* ctx, hmgpu_pic* out);
hmgpu_status hmgpu_picture_release(hmgpu_ctx* ctx, hmgpu_pic pic);
hmgpu_status hmgpu_picture_upload(hmgpu_ctx* ctx, hmgpu_pic pic, const int16_t* const planes[3], const int32_t strides[3]);
hmgpu_status hmgpu_picture_download(hmgpu_ctx* ctx, hmgpu_pic pic, int16_t* const planes[3], const int32_t strides[3]);

/* The same without waiting: the copies are enqueued and the call returns a ticket; hmgpu_download_wait(ticket) blocks until those
 * copies have landed.  hmgpu_download_wait may be called from another thread than the one that drives the context (a thread that
 * hashes or writes out pictures while the decoding thread keeps the device fed); planes should be page-locked (hmgpu_host_alloc). */
hmgpu_status hmgpu_picture_download_begin(hmgpu_ctx* ctx, hmgpu_pic pic, int16_t* const planes[3], const int32_t strides[3], uint64_t* ticket);
hmgpu_status hmgpu_download_wait(hmgpu_ctx* ctx, uint64_t ticket);

/* Output side.  hmgpu_picture_download_packed: the finished picture as the application wants it -- one or two
 * bytes per sample (TVideoIOYuv::write, TVideoIOYuv.cpp:706-790: 8-bit files take the low byte), cropped to a window given in
 * luma samples (conformance / display window; 0,0,0,0 = the whole picture) -- converted on the device, so an 8-bit picture crosses
 * PCIe in half the bytes.  hmgpu_picture_hash: the decoded-picture-hash SEI check (TDecGop.cpp:199-208) without moving the
 * picture: method 1 = MD5 (TComPicYuvMD5.cpp:183-205; HM's default and the hash its own streams carry), 2 = CRC, 3 = checksum
 * (:89-170).  MD5 is one serial chain of 64-byte blocks per plane -- about 0.2 s for a 3840x2160 10-bit luma plane on a GPU lane,
 * eight times what a host core needs -- so hmgpu_picture_hash(.., 1, ..) is for verification and tests.  A decoder uses
 * hmgpu_picture_hash_begin: it packs the planes behind the picture's filters (the picture buffer is free again at once) and the
 * chains of up to 32 pictures at a time run side by side, one LANE per plane, on a low-priority stream of their own; 48 bytes come
 * back per picture.  hmgpu_hash_wait(ticket, block = 0) polls, (block = 1) waits -- and launches the batch the ticket belongs to
 * if it is still collecting.  At most 96 tickets may be outstanding. */
hmgpu_status hmgpu_picture_download_packed(hmgpu_ctx* ctx, hmgpu_pic pic, void* const planes[3], const int32_t stride_bytes[3],
                                           int32_t bytes_per_sample, int32_t crop_left, int32_t crop_right, int32_t crop_top, int32_t crop_bottom);
hmgpu_status hmgpu_picture_hash(hmgpu_ctx* ctx, hmgpu_pic pic, int32_t method, uint8_t digest[3][16], int32_t* digest_len);
hmgpu_status hmgpu_picture_hash_begin(hmgpu_ctx* ctx, hmgpu_pic pic, int32_t method /* 1 */, uint64_t* ticket);
hmgpu_status hmgpu_hash_wait(hmgpu_ctx* ctx, uint64_t ticket, int32_t block, uint8_t digest[3][16], int32_t* digest_len, int32_t* ready);

/* Frame-parallel exchange (BASELINE config #5): a finished picture is ONE contiguous device region (three
 * planes, replicated margins included -- HM's TComPicYuv after extendPicBorder, TComPicYuv.cpp:89-100) that another GPU
 * needs before it can predict from the picture (TComPrediction.cpp:593).  The collective itself (RCCL broadcast /
 * send-recv between the owner's region and the receivers' regions) is issued by the caller on hmgpu_stream():
 *   sender:    hmgpu_picture_device_region(ctx, pic, HMGPU_REGION_FINISHED, &base, &bytes)   (extends the border if needed)
 *   receiver:  hmgpu_picture_device_region(ctx, pic, HMGPU_REGION_RECEIVE, &base, &bytes); <collective>;
 *              hmgpu_picture_commit_received(ctx, pic)        (the picture can now be named in ref_pic[][])
 * Regions of two contexts with equal hmgpu_seq_params have equal size and layout. */
typedef enum { HMGPU_REGION_FINISHED = 0, HMGPU_REGION_RECEIVE = 1 } hmgpu_region;
/* Within ONE process that drives several devices (a decoder that places the pictures of one temporal level on different GPUs,
 * TDecTop.cpp:672 / TDecGop.cpp:105 per picture): hmgpu_picture_transfer copies the finished picture `src_pic` of `src` into
 * `dst_pic` of `dst` -- a context of the same geometry on another GPU (peer copy over xGMI, hipMemcpyPeerAsync) or on the same one --
 * ordered behind everything enqueued on both contexts' streams, margins included, and commits it: dst_pic can be named in
 * ref_pic[][] of dst at once.  hmgpu_transfer_bytes: bytes a context has sent so far. */
hmgpu_status hmgpu_picture_transfer(hmgpu_ctx* src, hmgpu_pic src_pic, hmgpu_ctx* dst, hmgpu_pic dst_pic);
uint64_t     hmgpu_transfer_bytes(const hmgpu_ctx* ctx);
hmgpu_status hmgpu_picture_device_region(hmgpu_ctx* ctx, hmgpu_pic pic, int32_t which, void** base, int64_t* bytes);
hmgpu_status hmgpu_picture_commit_received(hmgpu_ctx* ctx, hmgpu_pic pic);
/* the context's HIP stream (a hipStream_t), for callers that order their own device work against the library's */
void* hmgpu_stream(hmgpu_ctx* ctx);

/* ------------------------------------------------------------------------------------------------ call 1
 * Replaces the reconstruction half of TDecGop::decompressSlice -> TDecSlice::decompressSlice ->
 * TDecCu::decompressCU (TDecSlice.cpp:334, TDecCu.cpp:142,373) for the CTUs [first_ctu, first_ctu+num_ctus) of
 * slice `slice_idx` of picture `cur`: motion compensation of every inter PU (TComPrediction::motionCompensation),
 * de-quantisation + inverse transform of every coded TU (TComTrQuant::invRecurTransformNxN) and
 * recon = ClipBD(pred + resid) into the picture (TComYuv::addClip, TDecCu::xCopyToPic).
 * Intra CUs (TDecCu::xReconIntraQT) are reconstructed too, in decoding order, when meta->intra_dir[] is supplied; without
 * the modes they are left untouched.  hmgpu_get_stats counts both kinds of partitions.
 * The metadata/coefficients are copied to the device before the call returns to the caller's thread?  No:
 * they are staged with hipMemcpyAsync from the caller's (ideally pinned) buffers, which must stay valid until
 * hmgpu_sync() or until a later call on the same context returns HMGPU_OK after a sync. */
hmgpu_status hmgpu_decompress_slice(hmgpu_ctx* ctx, hmgpu_pic cur, int32_t slice_idx, const hmgpu_slice_params* slice,
                                    const hmgpu_ctu_meta* meta, const hmgpu_coeffs* coeffs,
                                    int32_t first_ctu, int32_t num_ctus);

/* The same for a whole picture at once: every slice's constants first, then all CTUs in one batch of launches (meta->slice_idx
 * says which slice a CTU belongs to; required when num_slices > 1).  For callers that hold a complete parsed picture -- a decoder
 * that parses ahead of reconstruction, as libhmdec does -- and for pictures whose slices are not contiguous CTU ranges in raster
 * order (slices and tiles combined).  Equivalent to the hmgpu_decompress_slice calls of all slices. */
hmgpu_status hmgpu_decompress_picture(hmgpu_ctx* ctx, hmgpu_pic cur, int32_t num_slices, const hmgpu_slice_params* const* slices,
                                      const hmgpu_ctu_meta* meta, const hmgpu_coeffs* coeffs);

/* ------------------------------------------------------------------------------------------------ several pictures per call
 * TDecGop::decompressSlice / filterPicture (TDecGop.cpp:105,157) are called once per picture; a caller that holds several parsed
 * pictures which do not reference each other -- the B pictures of one temporal level, the pictures of independent streams -- hands
 * them over together: the inputs of all of them are staged on a copy stream of their own (so they travel while the kernels of the
 * previous call still run) and every kernel is launched ONCE for the whole set (n <= 16).  Equivalent to hmgpu_decompress_picture /
 * hmgpu_filter_picture per picture.  The copies are asynchronous: the input arrays must stay untouched until hmgpu_staging_wait()
 * (arrays of a staging block) or hmgpu_sync() (any arrays) has returned -- a later call that names the same picture only orders the
 * DEVICE side behind the earlier copy, it does not wait for it on the host. */
typedef struct hmgpu_picture_job {
  hmgpu_pic pic;
  int32_t num_slices;
  const hmgpu_slice_params* const* slices;
  const hmgpu_ctu_meta* meta;
  const hmgpu_coeffs* coeffs;
} hmgpu_picture_job;
typedef struct hmgpu_filter_job {
  hmgpu_pic pic;
  const hmgpu_pic_params* pp;
  const hmgpu_sao_param* sao;      /* [num_ctus][3], or NULL when pp->sao_enabled == 0 */
} hmgpu_filter_job;
hmgpu_status hmgpu_decompress_pictures(hmgpu_ctx* ctx, int32_t n, const hmgpu_picture_job* jobs);
hmgpu_status hmgpu_filter_pictures(hmgpu_ctx* ctx, int32_t n, const hmgpu_filter_job* jobs);

/* Staging blocks: ONE page-locked allocation that holds all input arrays of a picture (TComDataCU's arrays, TComDataCU.h:86-157, and
 * the levels m_pcTrCoeff*) in the order the device keeps them.  hmgpu_staging_alloc fills *meta and *coeffs with pointers into the
 * block -- a parser writes HM's arrays there directly, part_size pre-set to HMGPU_SIZE_NONE, ref_idx to -1, everything else to 0 --
 * and a whole-picture call that is handed exactly these structs moves the metadata in one DMA and the levels in another instead of
 * one copy per array.  PCM samples are not part of the block (coeffs->pcm_sample stays the caller's). */
typedef struct hmgpu_staging hmgpu_staging;
hmgpu_status hmgpu_staging_alloc(hmgpu_ctx* ctx, hmgpu_staging** out, hmgpu_ctu_meta* meta, hmgpu_coeffs* coeffs);
void         hmgpu_staging_free(hmgpu_ctx* ctx, hmgpu_staging* staging);
/* blocks until the copies of the last hmgpu_decompress_pictures call that read the block have been made: from then on a parser may
 * write the next picture into it (the kernels read the device copies).  A block no call has read yet returns at once. */
hmgpu_status hmgpu_staging_wait(hmgpu_ctx* ctx, hmgpu_staging* staging);
/* A caller that drives several contexts (one per GPU, pictures of one temporal level placed on different ones: hmgpu_picture_transfer)
 * parses into ONE set of blocks and decides late where a picture is decoded: after hmgpu_staging_share, `other` -- a context with equal
 * hmgpu_seq_params on any device -- takes the block's arrays in the same few DMAs as `owner`, which still owns and frees the block. */
hmgpu_status hmgpu_staging_share(hmgpu_ctx* owner, hmgpu_staging* staging, hmgpu_ctx* other);
/* (the block also holds the three ctu_level_start arrays: coeffs->ctu_level_start[] of hmgpu_staging_alloc points at them; a caller
 * that fills the block with HM's dense layout sets the three pointers to NULL in the struct it passes to the calls) */

/* HM's dense level arrays -> the compact form (host, no device involved): out_level[c] needs room for the dense size in the worst
 * case, out_start[c] for num_ctus + 1 entries.  The walk over depth / tr_idx / cbf is the one the device's flattener does. */
hmgpu_status hmgpu_pack_levels(const hmgpu_seq_params* seq, const hmgpu_ctu_meta* meta, const hmgpu_coeffs* dense,
                               int16_t* const out_level[3], uint32_t* const out_start[3]);

/* ------------------------------------------------------------------------------------------------ call 2
 * Replaces TDecGop::filterPicture (TDecGop.cpp:157-217): TComLoopFilter::loopFilterPic (all vertical edges, then
 * all horizontal edges), then reconstructBlkSAOParams + SAOProcess.  Uses the metadata of every CTU handed to
 * hmgpu_decompress_slice for this picture (HM: pcPic->getCU(addr), TComLoopFilter.cpp:133-135).
 * `sao` is [num_ctus][3] or NULL when pp->sao_enabled == 0. */
hmgpu_status hmgpu_filter_picture(hmgpu_ctx* ctx, hmgpu_pic cur, const hmgpu_pic_params* pp, const hmgpu_sao_param* sao);

/* ------------------------------------------------------------------------------------------------ finer seams
 * The kernel-level seams HM exposes to its own callers (SURVEY.md 8b "finer seams"), used by the parity tests.
 * All operate on host arrays (copied in and out synchronously). */

/* TComTrQuant::xDeQuant (flat) + xIT/xITransformSkip on `n` TUs of size (1<<log2_size)^2.
 * levels/resid: [n][size*size].  per TU: qp_per/qp_rem from QpParam, flags bit0 = 4x4 DST (intra luma), bit1 = transform skip.
 * bit_depth selects transformShift and the second-stage shift (TComTrQuant.cpp:898-899,1233-1236). */
hmgpu_status hmgpu_inverse_transform_batch(hmgpu_ctx* ctx, int32_t log2_size, int32_t bit_depth, int32_t n,
                                           const int16_t* levels, const int8_t* qp_per, const int8_t* qp_rem,
                                           const uint8_t* flags, int16_t* resid);

/* TComPrediction::xPredInterBlk (TComPrediction.cpp:660) for `n` blocks out of one reference plane.
 * blocks: n x {x, y, w, h, mvx, mvy} in samples of that plane / in 1/4 (luma) or 1/8 (chroma) sample units;
 * dst: concatenated w*h outputs.  bi != 0 -> 14-bit intermediate (no clip), else final clipped prediction. */
hmgpu_status hmgpu_mc_batch(hmgpu_ctx* ctx, int32_t is_chroma, int32_t bit_depth, const int16_t* ref_plane, int32_t ref_stride,
                            int32_t ref_w, int32_t ref_h, int32_t n, const int32_t* blocks, int32_t bi, int16_t* dst);

/* stage control for tests: run only part of hmgpu_filter_picture on `cur`:  1 = vertical edges, 2 = horizontal edges,
 * 4 = SAO; combine with |.  hmgpu_filter_picture == stages 7. */
hmgpu_status hmgpu_filter_picture_stages(hmgpu_ctx* ctx, hmgpu_pic cur, const hmgpu_pic_params* pp, const hmgpu_sao_param* sao,
                                         int32_t stages);

/* ------------------------------------------------------------------------------------------------ measurement
 * Resident replay: re-run the device work of the last hmgpu_decompress_slice calls / hmgpu_filter_picture of `cur`
 * from the inputs already staged in HBM (no host->device traffic, no host work), `iters` times.  This is what
 * bench.py times ("inputs already resident in HBM").  Kernel times are measured with hipEvents on the
 * context's own stream. */
hmgpu_status hmgpu_replay(hmgpu_ctx* ctx, hmgpu_pic cur, int32_t stages /* 8 = reconstruct | 1|2|4 filter */, int32_t iters);
/* the same for `n` mutually independent pictures at once (<= 16): every kernel is launched once for the whole batch
 * (one grid z-slice per picture) -- the frame-parallel mode used for throughput measurements */
hmgpu_status hmgpu_replay_batch(hmgpu_ctx* ctx, const hmgpu_pic* pics, int32_t n, int32_t stages, int32_t iters);

/* Lanes of hmgpu_replay_batch: 1 (default) = one stream, kernel after kernel; 2 = the batch is cut in two halves that run on two
 * streams, so that kernels of different kinds overlap (more pictures per second, per-kernel times no longer separable; profiling
 * forces one lane). */
hmgpu_status hmgpu_set_streams(hmgpu_ctx* ctx, int32_t n);

#define HMGPU_NUM_KERNELS 12
typedef struct hmgpu_stats {
  double   kernel_ms[HMGPU_NUM_KERNELS];      /* accumulated device time per kernel class since the last reset */
  uint64_t kernel_launches[HMGPU_NUM_KERNELS];
  uint64_t intra_partitions;                  /* 4x4 partitions of intra CUs seen */
  uint64_t inter_partitions;
  uint64_t coded_tus[4][3];                   /* TUs with cbf by log2 size-2 and component */
} hmgpu_stats;
const char*  hmgpu_kernel_name(int32_t k);
hmgpu_status hmgpu_set_profiling(hmgpu_ctx* ctx, int32_t enable);   /* per-kernel hipEvent timing on/off (off by default) */
hmgpu_status hmgpu_get_stats(hmgpu_ctx* ctx, hmgpu_stats* out, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* HMGPU_H */
