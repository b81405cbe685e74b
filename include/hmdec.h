/* hmdec.h -- C interface of libhmdec.so: a complete HEVC (Main / Main10) decoder whose pixel path is libhmgpu (MI355X) and whose
 * bitstream side is a from-scratch host parser.
 *
 * The first block below is call-compatible with ChristianFeldmann/libHM's libHMDecoder (source/App/libHMDecoder/libHMDecoder.h:111-298):
 * same names, argument lists, enumerator values and protocol, so a client built against that header links against libhmdec.so
 * unchanged.  Each declaration cites the function it stands in for.  The protocol in short:
 *
 *   ctx = libHMDec_new_decoder();
 *   for every NAL unit:  libHMDec_push_nal_unit(ctx, data, len, eof, newPicture, checkOutput);
 *                        if (newPicture)  push the same unit again after draining the output
 *                        if (checkOutput) while ((pic = libHMDec_get_picture(ctx))) { ... libHMDEC_get_image_plane(pic, c) ... }
 *   libHMDec_free_decoder(ctx);
 *
 * The second block (hmdec_*) is this library's own: device selection, a parse-only mode for hosts without a GPU, and read access
 * to what the parser produced (the arrays that cross include/hmgpu.h), which the parity tests compare with HM's.
 */
#ifndef HMDEC_H
#define HMDEC_H

#include <stdint.h>

#ifdef __cplusplus
#include <vector>
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------- libHMDecoder-compatible part */
typedef enum { LIBHMDEC_OK = 0, LIBHMDEC_ERROR, LIBHMDEC_ERROR_READ_ERROR } libHMDec_error;            /* libHMDecoder.h:99-104 */
typedef void libHMDec_context;
typedef void libHMDec_picture;
typedef enum { LIBHMDEC_LUMA = 0, LIBHMDEC_CHROMA_U, LIBHMDEC_CHROMA_V } libHMDec_ColorComponent;     /* :166-171 */
typedef enum { LIBHMDEC_CHROMA_400 = 0, LIBHMDEC_CHROMA_420, LIBHMDEC_CHROMA_422, LIBHMDEC_CHROMA_444, LIBHMDEC_CHROMA_UNKNOWN } libHMDec_ChromaFormat; /* :222-229 */

const char* libHMDec_get_version(void);                                       /* :108  "16.0": the HM version whose output is matched */
libHMDec_context* libHMDec_new_decoder(void);                                 /* :119 */
libHMDec_error libHMDec_free_decoder(libHMDec_context* decCtx);               /* :125 */
void libHMDec_set_SEI_Check(libHMDec_context* decCtx, bool check_hash);       /* :132  decoded-picture-hash SEI check on/off (default on) */
void libHMDec_set_max_temporal_layer(libHMDec_context* decCtx, int max_layer);/* :139  -1 = all layers */
#ifdef __cplusplus
/* :152  one NAL unit, start code optional.  bNewPicture: the unit opened a new picture, the finished one may now be read, and the
 * unit must be pushed again.  checkOutputPictures: call libHMDec_get_picture until it returns NULL. */
libHMDec_error libHMDec_push_nal_unit(libHMDec_context* decCtx, const void* data8, int length, bool eof, bool& bNewPicture, bool& checkOutputPictures);
#endif
libHMDec_picture* libHMDec_get_picture(libHMDec_context* decCtx);             /* :180  next picture in output order or NULL */
int libHMDEC_get_POC(libHMDec_picture* pic);                                  /* :188 */
int libHMDEC_get_picture_width(libHMDec_picture* pic, libHMDec_ColorComponent c);   /* :196  coded size, conformance window not applied */
int libHMDEC_get_picture_height(libHMDec_picture* pic, libHMDec_ColorComponent c);  /* :200 */
int libHMDEC_get_picture_stride(libHMDec_picture* pic, libHMDec_ColorComponent c);  /* :208  in samples */
short* libHMDEC_get_image_plane(libHMDec_picture* pic, libHMDec_ColorComponent c);  /* :217  valid until the picture buffer is reused */
libHMDec_ChromaFormat libHMDEC_get_chroma_format(libHMDec_picture* pic);      /* :235 */
int libHMDEC_get_internal_bit_depth(libHMDec_ColorComponent c);               /* :241  of the most recently activated SPS (HM: a global) */

typedef struct { unsigned short x, y, w, h; int value; int value2; } libHMDec_BlockValue;   /* :248-253 */
typedef enum {                                                                /* :257-282 */
  LIBHMDEC_CTU_SLICE_INDEX = 0, LIBHMDEC_CU_PREDICTION_MODE, LIBHMDEC_CU_TRQ_BYPASS, LIBHMDEC_CU_SKIP_FLAG, LIBHMDEC_CU_PART_MODE,
  LIBHMDEC_CU_INTRA_MODE_LUMA, LIBHMDEC_CU_INTRA_MODE_CHROMA, LIBHMDEC_CU_ROOT_CBF, LIBHMDEC_PU_MERGE_FLAG, LIBHMDEC_PU_MERGE_INDEX,
  LIBHMDEC_PU_UNI_BI_PREDICTION, LIBHMDEC_PU_REFERENCE_POC_0, LIBHMDEC_PU_MV_0, LIBHMDEC_PU_REFERENCE_POC_1, LIBHMDEC_PU_MV_1,
  LIBHMDEC_TU_CBF_Y, LIBHMDEC_TU_CBF_CB, LIBHMDEC_TU_CBF_CR, LIBHMDEC_TU_COEFF_TR_SKIP_Y, LIBHMDEC_TU_COEFF_TR_SKIP_Cb,
  LIBHMDEC_TU_COEFF_TR_SKIP_Cr, LIBHMDEC_TU_COEFF_ENERGY_Y, LIBHMDEC_TU_COEFF_ENERGY_CB, LIBHMDEC_TU_COEFF_ENERGY_CR
} libHMDec_info_type;
#ifdef __cplusplus
std::vector<libHMDec_BlockValue>* libHMDEC_get_internal_info(libHMDec_context* decCtx, libHMDec_picture* pic, libHMDec_info_type type);  /* :293 */
#endif
libHMDec_error libHMDEC_clear_internal_info(libHMDec_context* decCtx);        /* :300 */

/* ---------------------------------------------------------------------------------------------- this library's own additions */
void hmdec_set_device(libHMDec_context* ctx, int device_ordinal);             /* GPU to decode on (default 0); before the first NAL unit */
/* Several device contexts, one per entry (GPU ordinals; the same ordinal twice: two contexts on one GPU).  The pictures libHM would
 * reconstruct one after the other and that do not predict from each other -- the B pictures of one temporal level, TDecTop.cpp:672 /
 * TDecGop.cpp:105 per picture -- are placed round-robin on the contexts; a reference picture is copied once to each context that
 * predicts from it (hmgpu_picture_transfer: over xGMI between GPUs), reference picture sets and lists (TComSlice.cpp:318-376), hash
 * checks and the output order (TDecTop.cpp:192-213) are as with one.  Before the first NAL unit.  hmdec_transfer_bytes: bytes copied. */
void hmdec_set_devices(libHMDec_context* ctx, const int* device_ordinals, int n);
int hmdec_num_devices(libHMDec_context* ctx);
unsigned long long hmdec_transfer_bytes(libHMDec_context* ctx);
/* parser threads (1..16, default 1): slice data of several pictures is parsed concurrently (frame-parallel, a picture at most one
 * CTB row behind the picture it takes temporal motion vectors from); pictures come out in the same order, later.  Before the first NAL unit. */
void hmdec_set_threads(libHMDec_context* ctx, int n);
void hmdec_set_parse_only(libHMDec_context* ctx, int on);                     /* no device work: parser output only (planes unavailable) */
int hmdec_hash_mismatches(libHMDec_context* ctx);                             /* pictures whose reconstruction disagreed with the hash SEI */
int hmdec_pictures_decoded(libHMDec_context* ctx);
void hmdec_set_device_md5(libHMDec_context* ctx, int on);                     /* MD5 hash SEIs checked on the device (hmgpu_picture_hash_begin) (the default;
                                                                                  HMDEC_DEVICE_MD5=0 or 0 here: on the decoder's hash threads) */
int hmdec_device_batches(libHMDec_context* ctx);                              /* calls of hmgpu_decompress_pictures so far (pictures retired together share one) */
const char* hmdec_last_error(libHMDec_context* ctx);
libHMDec_picture* hmdec_last_decoded_picture(libHMDec_context* ctx);          /* the picture finished most recently, decoding order */
libHMDec_picture* hmdec_open_picture(libHMDec_context* ctx);                  /* the picture whose slices are still arriving, or NULL */
/* parser output of a picture by name: "depth", "part_size", "pred_mode", "qp", "tr_idx", "cbf0".."cbf2", "ts0".."ts2", "mv0", "mv1",
 * "ref_idx0", "ref_idx1", "intra_dir0", "intra_dir1", "bypass", "ipcm", "skip", "merge", "slice_idx", "tile_idx", "coeff0".."coeff2",
 * "pcm0".."pcm2", "sao", "plane0".."plane2" -- the HM-layout arrays of include/hmgpu.h.  Returns 0 and pointer/size on success. */
int hmdec_picture_array(libHMDec_picture* pic, const char* name, const void** data, int64_t* bytes);
int hmdec_picture_num_slices(libHMDec_picture* pic);
int hmdec_picture_slice_params(libHMDec_picture* pic, int slice, void* out /* hmgpu_slice_params */, void* lists_out /* hmgpu_scaling_lists or NULL */);
/* geometry and sequence-level constants of a picture: width, height, log2 CTB size, bit depth luma, bit depth chroma, PCM bit depth
 * luma, PCM bit depth chroma, pcm_loop_filter_disabled (and PCM enabled), strong_intra_smoothing, SAO enabled, loop filter across
 * tiles, number of CTBs -- what hmgpu_seq_params / hmgpu_pic_params are filled from */
int hmdec_picture_geometry(libHMDec_picture* pic, int32_t out[12]);
/* HMGPU_REXT_* of the picture's SPS (hmgpu_seq_params.range_ext_flags) */
int hmdec_picture_range_ext_flags(libHMDec_picture* pic);
/* chroma_format_idc of the picture's SPS (hmgpu_seq_params.chroma_format): 0 4:0:0, 1 4:2:0, 2 4:2:2, 3 4:4:4 */
int hmdec_picture_chroma_format(libHMDec_picture* pic);
/* PPS log2_sao_offset_scale_luma / _chroma of the picture (hmgpu_pic_params.sao_offset_shift_*) */
int hmdec_picture_sao_offset_shift(libHMDec_picture* pic, int chroma);
/* conformance window of the picture's SPS in luma samples: left, right, top, bottom (libHM hands out the uncropped picture) */
int hmdec_picture_conformance_window(libHMDec_picture* pic, int32_t window[4]);
/* libHMDEC_get_internal_info for C callers: pointer to the first element and the count (same storage, same lifetime) */
int hmdec_internal_info(libHMDec_context* ctx, libHMDec_picture* pic, int type, const libHMDec_BlockValue** data);
int hmdec_picture_hash_sei(libHMDec_picture* pic, uint8_t digest[48]);        /* returns the method (0 none, 1 MD5, 2 CRC, 3 checksum) */

#ifdef __cplusplus
}
#endif
#endif
