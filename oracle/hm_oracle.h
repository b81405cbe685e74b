/* oracle/hm_oracle.h -- TEST INFRASTRUCTURE ONLY (never linked, loaded or called by the product).
 *
 * Plain-C, single-threaded restatement of the HM 16.0 decoder's pixel-reconstruction path, function by
 * function in HM's own (serial, recursive) structure, operating on the same input structs as the product's
 * C ABI (include/hmgpu.h) so that tests can feed both from one set of arrays.
 * Pinned against HM itself: tests/test_oracle_*.py check every function below against the golden fixtures
 * under tests/golden/ (made by oracle/make_golden.py from the real HM libraries) and, when
 * oracle/_ref/libhmref.so is present, against live calls into HM.
 */
#ifndef HM_ORACLE_H
#define HM_ORACLE_H
#include "../include/hmgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* dense picture: three planes, stride = plane width */
typedef struct hmo_picture {
  int16_t* plane[3];
} hmo_picture;

/* ---- kernel level ---- */
/* xDeQuant flat path (TComTrQuant.cpp:1276-1311): n = size*size levels */
void hmo_dequant(const int16_t* level, int32_t* coef, int n, int log2_size, int bit_depth, int qp_per, int qp_rem);
/* xITrMxN (TComTrQuant.cpp:894-948): square N x N, int32 coefficient in, int32 residual out */
void hmo_itr(int bit_depth, const int32_t* coeff, int32_t* block, int n, int use_dst);
/* dequant + (IT | transform skip) of one TU: invTransformNxN (TComTrQuant.cpp:1423-1548), flags bit0 DST, bit1 skip */
void hmo_inverse_transform_tu(const int16_t* level, int16_t* resid, int resid_stride, int log2_size, int bit_depth,
                              int qp_per, int qp_rem, int flags);
/* the same with scaling lists (NULL = flat): list_id = 3 * inter + component (getScalingListType) */
void hmo_inverse_transform_tu_sl(const int16_t* level, int16_t* resid, int resid_stride, int log2_size, int bit_depth,
                                 int qp_per, int qp_rem, int flags, const hmgpu_scaling_lists* sl, int list_id);
/* rotation + RDPCM of a block that skipped the transform (TComTrQuant.cpp:1475-1487, 1737-1792): rdpcm 0 off, 1 hor, 2 ver */
void hmo_residual_rotate_rdpcm(int16_t* resid, int stride, int n, int rotate, int rdpcm);
/* picture hashes of one plane (TComPicYuvMD5.cpp:89-170): CRC-16 (2 bytes, big endian in out) and checksum (4 bytes) */
void hmo_plane_crc(int bit_depth, const int16_t* plane, int width, int height, int stride, uint8_t out[2]);
void hmo_plane_checksum(int bit_depth, const int16_t* plane, int width, int height, int stride, uint8_t out[4]);
/* QpParam (TComTrQuant.cpp:71-100): comp 0..2 */
void hmo_qp_param(int qp_y, int comp, int bit_depth, int chroma_qp_offset, int* per, int* rem);
/* xPredInterBlk (TComPrediction.cpp:660-698) with the reference addressed by clamped coordinates.
 * (bx,by) block origin in the plane, mv in 1/4 (luma) or 1/8 (chroma) sample units */
void hmo_pred_inter_blk(int is_chroma, int bit_depth, const int16_t* ref, int ref_stride, int ref_w, int ref_h,
                        int bx, int by, int w, int h, int mvx, int mvy, int bi, int16_t* dst, int dst_stride);
/* TComYuv::addAvg (TComYuv.cpp:336) */
void hmo_add_avg(const int16_t* s0, const int16_t* s1, int16_t* dst, int w, int h, int stride, int bit_depth);
/* offsetBlock (TComSampleAdaptiveOffset.cpp:375-661); avail[8] = L,R,A,B,AL,AR,BL,BR */
void hmo_sao_offset_block(int bit_depth, int type_idx, const int32_t* offset, const int16_t* src, int16_t* res,
                          int src_stride, int res_stride, int w, int h, const int32_t* avail);

/* ---- picture level ---- */
/* TDecSlice/TDecCu::decompressCU over all CTUs [first_ctu, first_ctu+num_ctus).  refs[handle] are the reference
 * pictures (dense).  intra CUs: skipped (pixels of `cur` left as they are) -- counts returned in *n_intra_parts. */
int hmo_decompress_ctus(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                        const hmgpu_coeffs* coeffs, hmo_picture* cur, const hmo_picture* refs, int num_refs,
                        int first_ctu, int num_ctus, int64_t* n_intra_parts);
/* TComLoopFilter::loopFilterPic (TComLoopFilter.cpp:130-155), in place; dir_mask bit0 vertical edges, bit1 horizontal */
int hmo_loop_filter_pic(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                        const hmgpu_pic_params* pp, hmo_picture* pic, int dir_mask);
/* boundary strengths the way xDeblockCU leaves them: bs[dir][num_ctus*parts] (for tests of the GPU Bs derivation) */
int hmo_boundary_strengths(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                           const hmgpu_pic_params* pp, uint8_t* bs_ver, uint8_t* bs_hor);
/* reconstructBlkSAOParams (TComSampleAdaptiveOffset.cpp:348-372): raw -> reconstructed, [num_ctus][3] */
int hmo_sao_reconstruct_params(const hmgpu_seq_params* seq, const hmgpu_pic_params* pp, const hmgpu_ctu_meta* meta,
                               const hmgpu_sao_param* raw, hmgpu_sao_param* rec);
/* SAOProcess (TComSampleAdaptiveOffset.cpp:717-734): src (deblocked) -> dst; dst must start as a copy of src */
int hmo_sao_process(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_pic_params* pp,
                    const hmgpu_ctu_meta* meta, const hmgpu_sao_param* rec, const hmo_picture* src, hmo_picture* dst);
/* per-plane MD5 the way TComPicYuvMD5.cpp:183-205 feeds libmd5: out 3 x 16 bytes */
void hmo_picture_md5(const hmgpu_seq_params* seq, const hmo_picture* pic, uint8_t* out48);

#ifdef __cplusplus
}
#endif
#endif
