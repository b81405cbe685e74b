/* oracle/hm_oracle.c -- TEST INFRASTRUCTURE ONLY (see hm_oracle.h).
 *
 * Serial plain-C restatement of the HM 16.0 decoder pixel path.  It deliberately keeps HM's structure
 * (recursive CU/TU descent, per-CTU edge/Bs arrays, CTU-by-CTU SAO) so that each function can be read
 * next to the HM function it follows (file:line in every comment, relative to
 * /root/reference/source/Lib/).  Nothing here is shared with the product (libhm_amd/), which is
 * data-parallel and organised completely differently.
 */
#include "hm_oracle.h"
#include <stdlib.h>
#include <string.h>

#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
static int iabs(int v) { return v < 0 ? -v : v; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------------ ROM */
/* TLibCommon/TComRom.cpp:335-417 (DCT), :456-484 (DST): the 32-point matrix sampled at the odd multiples of
 * pi/64; every smaller DCT is a row-subsampled copy (TComRom.cpp DEFINE_DCT*_MATRIX). */
static const int k_cos64[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
static int dct_coef(int n_size, int k, int n)    /* g_aiT<N>[TRANSFORM_INVERSE][k][n] */
{
  int a = ((2 * n + 1) * k * (32 / n_size)) & 127;
  if (a <= 32) return k_cos64[a];
  if (a <= 64) return -k_cos64[64 - a];
  if (a <= 96) return -k_cos64[a - 64];
  return k_cos64[128 - a];
}
static const int k_dst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
static const int k_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                      /* TComRom.cpp:326-329 */
static const unsigned char k_chroma_scale_420[58] = {                                     /* TComRom.cpp:503 */
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
  33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
static const unsigned char k_tc_table[54] = {                                             /* TComLoopFilter.cpp:59 */
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10,
  11, 13, 14, 16, 18, 20, 22, 24 };
static const unsigned char k_beta_table[52] = {                                           /* TComLoopFilter.cpp:64 */
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38,
  40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
static const int k_luma_filter[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                         { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int k_chroma_filter[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                           { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

/* ------------------------------------------------------------------------------------------------ de-quantisation */
/* TComTrQuant::xDeQuant, flat branch: TLibCommon/TComTrQuant.cpp:1276-1311 */
void hmo_dequant(const int16_t* level, int32_t* coef, int n, int log2_size, int bit_depth, int qp_per, int qp_rem)
{
  const int transform_shift = 15 - bit_depth - log2_size;          /* getTransformShift, TComChromaFormat.h:166 */
  const int right_shift = 6 - (transform_shift + qp_per);          /* IQUANT_SHIFT = 6 */
  const int scale = k_inv_quant_scales[qp_rem];
  int target_bits = 32 + right_shift - 7;                          /* min(maxTrDynamicRange+1, 32+rightShift-scaleBits) */
  if (target_bits > 16) target_bits = 16;
  {
    const int in_min = -(1 << (target_bits - 1)), in_max = (1 << (target_bits - 1)) - 1;
    int i;
    if (right_shift > 0)
    {
      const int add = 1 << (right_shift - 1);
      for (i = 0; i < n; i++)
      {
        int q = CLIP3(in_min, in_max, (int)level[i]);
        int c = (q * scale + add) >> right_shift;
        coef[i] = CLIP3(-32768, 32767, c);
      }
    }
    else
    {
      const int left_shift = -right_shift;
      for (i = 0; i < n; i++)
      {
        int q = CLIP3(in_min, in_max, (int)level[i]);
        int c = (int)((unsigned)(q * scale) << left_shift);   /* HM: signed << ; value stays inside int32 for bit depth <= 10 */
        coef[i] = CLIP3(-32768, 32767, c);
      }
    }
  }
}

/* getScaledChromaQP: g_aucChromaScale[chFmt] (TComRom.cpp:499-506, TComChromaFormat.h:178-181): the 4:2:0 table, min(qPi, 51) otherwise */
static int scaled_chroma_qp(int qp, int chroma_format)
{
  qp = CLIP3(0, 57, qp);
  return chroma_format == 1 ? k_chroma_scale_420[qp] : imin(qp, 51);
}
/* QpParam::QpParam: TLibCommon/TComTrQuant.cpp:71-100 */
void hmo_qp_param_fmt(int qp_y, int comp, int bit_depth, int chroma_qp_offset, int chroma_format, int* per, int* rem)
{
  const int qp_bd_offset = 6 * (bit_depth - 8);
  int base;
  if (comp == 0) base = qp_y + qp_bd_offset;
  else
  {
    base = CLIP3(-qp_bd_offset, 57, qp_y + chroma_qp_offset);
    if (base < 0) base = base + qp_bd_offset;
    else base = scaled_chroma_qp(base, chroma_format) + qp_bd_offset;
  }
  *per = base / 6; *rem = base % 6;
}
void hmo_qp_param(int qp_y, int comp, int bit_depth, int chroma_qp_offset, int* per, int* rem)
{
  hmo_qp_param_fmt(qp_y, comp, bit_depth, chroma_qp_offset, 1, per, rem);
}

/* ------------------------------------------------------------------------------------------------ inverse transform */
/* transform matrices as tables: g_tmat[log2N-2][k][n] = g_aiT<N>[TRANSFORM_INVERSE][k][n] */
static int g_tmat[4][32][32];
static int g_tmat_ready = 0;
static void tmat_init(void)
{
  int l, k, n;
  if (g_tmat_ready) return;
  for (l = 0; l < 4; l++) for (k = 0; k < (4 << l); k++) for (n = 0; n < (4 << l); n++) g_tmat[l][k][n] = dct_coef(4 << l, k, n);
  g_tmat_ready = 1;
}
/* filled when the library is loaded, so that concurrent callers (bench.py's all-cores leg) only ever read the tables */
__attribute__((constructor)) static void tmat_ctor(void) { tmat_init(); }

/* N-point inverse DCT of one column by even/odd decomposition, the structure of partialButterflyInverse4/8/16/32
 * (TComTrQuant.cpp:468-828): O[k] from the odd rows, E[] = the N/2-point transform of the even rows, no rounding
 * until the very end.  in[m] = src[m*stride]; out[k], k = 0..n-1, unrounded. */
static void idct_1d(int log2n, const int32_t* src, int stride, int* out)
{
  const int n = 1 << log2n;
  if (n == 2)
  {
    /* rows 0 and 1 of the 2-point kernel: 64, 64 / 64, -64 */
    const int a = 64 * src[0], b = 64 * src[stride];
    out[0] = a + b; out[1] = a - b;
    return;
  }
  {
    int e[16], o[16], k, m;
    const int (*t)[32] = g_tmat[log2n - 2];
    idct_1d(log2n - 1, src, stride * 2, e);
    for (k = 0; k < n / 2; k++)
    {
      int sum = 0;
      for (m = 1; m < n; m += 2) sum += t[m][k] * src[m * stride];
      o[k] = sum;
    }
    for (k = 0; k < n / 2; k++) { out[k] = e[k] + o[k]; out[n - 1 - k] = e[k] - o[k]; }
  }
}

/* one 1-D stage exactly as partialButterflyInverseN is used by xITrMxN: reads column j of src (stride n), writes the
 * n results contiguously at dst[j*n ..] (i.e. transposed). */
static void inv_stage(const int32_t* src, int32_t* dst, int n, int shift, int lo, int hi, int use_dst)
{
  const int add = (shift > 0) ? (1 << (shift - 1)) : 0;
  int j, k, m;
  int log2n = 2;
  while ((1 << log2n) < n) log2n++;
  tmat_init();
  for (j = 0; j < n; j++)
  {
    int out[32];
    if (use_dst)
    {
      /* fastInverseDst: TComTrQuant.cpp:437-462 */
      for (k = 0; k < 4; k++) { int sum = 0; for (m = 0; m < 4; m++) sum += k_dst4[m][k] * src[m * 4 + j]; out[k] = sum; }
    }
    else idct_1d(log2n, src + j, n, out);
    for (k = 0; k < n; k++)
    {
      const int v = (out[k] + add) >> shift;
      dst[j * n + k] = CLIP3(lo, hi, v);
    }
  }
}

/* xITrMxN: TLibCommon/TComTrQuant.cpp:894-948 (square blocks, maxTrDynamicRange 15) */
void hmo_itr(int bit_depth, const int32_t* coeff, int32_t* block, int n, int use_dst)
{
  int32_t tmp[32 * 32];
  const int shift_1st = 7;                 /* TRANSFORM_MATRIX_SHIFT(6) + 1 */
  const int shift_2nd = 20 - bit_depth;    /* 6 + 15 - 1 - bitDepth */
  const int dst = use_dst && n == 4;
  inv_stage(coeff, tmp, n, shift_1st, -32768, 32767, dst);
  inv_stage(tmp, block, n, shift_2nd, -32768, 32767, dst);   /* second clip: numeric_limits<Pel> */
}

/* invTransformNxN: TLibCommon/TComTrQuant.cpp:1423-1548 (no bypass, no RDPCM, no rotation) */
/* xDeQuant with scaling lists (TComTrQuant.cpp:1238-1275): per-position factor invQuantScale[rem] * m, m from the list of this
 * size and type, replicated over ratio x ratio positions for 16x16 / 32x32 with the DC entry at position 0
 * (xSetScalingListDec / processScalingListDec, :2992-3012, 3092-3106); four more fractional bits than the flat path */
static void dequant_lists(const int16_t* level, int32_t* coef, int log2_size, int bit_depth, int qp_per, int qp_rem,
                          const hmgpu_scaling_lists* sl, int list_id)
{
  const int n = 1 << log2_size, size_id = log2_size - 2;
  const int ratio = n > 8 ? n / 8 : 1, mn = n > 8 ? 8 : n;
  const int transform_shift = 15 - bit_depth - log2_size;
  const int right_shift = 6 - (transform_shift + qp_per) + 4;      /* + LOG2_SCALING_LIST_NEUTRAL_VALUE */
  const int scale = k_inv_quant_scales[qp_rem];
  int target_bits = 32 + right_shift - 15;                         /* dequantCoefBits = 1 + IQUANT_SHIFT + SCALING_LIST_BITS */
  int x, y;
  if (target_bits > 16) target_bits = 16;
  {
    const int in_min = -(1 << (target_bits - 1)), in_max = (1 << (target_bits - 1)) - 1;
    for (y = 0; y < n; y++)
      for (x = 0; x < n; x++)
      {
        const int m = (ratio > 1 && x == 0 && y == 0) ? sl->dc[size_id][list_id] : sl->coef[size_id][list_id][mn * (y / ratio) + x / ratio];
        const int q = CLIP3(in_min, in_max, (int)level[y * n + x]);
        int c;
        if (right_shift > 0) c = (q * (scale * m) + (1 << (right_shift - 1))) >> right_shift;
        else c = (int)((unsigned)(q * (scale * m)) << (-right_shift));
        coef[y * n + x] = CLIP3(-32768, 32767, c);
      }
  }
}

void hmo_inverse_transform_tu(const int16_t* level, int16_t* resid, int resid_stride, int log2_size, int bit_depth,
                              int qp_per, int qp_rem, int flags)
{
  hmo_inverse_transform_tu_sl(level, resid, resid_stride, log2_size, bit_depth, qp_per, qp_rem, flags, NULL, 0);
}

void hmo_inverse_transform_tu_sl(const int16_t* level, int16_t* resid, int resid_stride, int log2_size, int bit_depth,
                                 int qp_per, int qp_rem, int flags, const hmgpu_scaling_lists* sl, int list_id)
{
  const int n = 1 << log2_size;
  int32_t coef[32 * 32] = { 0 }, block[32 * 32];
  int x, y;
  /* getUseScalingList (TComTrQuant.h:180): not for transform-skip blocks other than 4x4 */
  if (sl && (!(flags & 2) || log2_size == 2)) dequant_lists(level, coef, log2_size, bit_depth, qp_per, qp_rem, sl, list_id);
  else hmo_dequant(level, coef, n * n, log2_size, bit_depth, qp_per, qp_rem);
  if (flags & 2)
  {
    /* xITransformSkip: TComTrQuant.cpp:1920-1959 */
    int tshift = 15 - bit_depth - log2_size;
    if (tshift >= 0)
    {
      const int offset = tshift == 0 ? 0 : (1 << (tshift - 1));
      for (y = 0; y < n; y++) for (x = 0; x < n; x++) resid[y * resid_stride + x] = (int16_t)((coef[y * n + x] + offset) >> tshift);
    }
    else
    {
      tshift = -tshift;
      for (y = 0; y < n; y++) for (x = 0; x < n; x++) resid[y * resid_stride + x] = (int16_t)(coef[y * n + x] << tshift);
    }
    return;
  }
  hmo_itr(bit_depth, coef, block, n, (flags & 1) != 0);
  for (y = 0; y < n; y++) for (x = 0; x < n; x++) resid[y * resid_stride + x] = (int16_t)block[y * n + x];   /* xIT: :1861-1865 */
}

/* What sps_range_extension() adds to a block that skipped the transform (transform-skip or cu_transquant_bypass):
 * rotation = the block read back to front (TComTrQuant.cpp:1475-1487 for bypass, :1943 inside xITransformSkip -- element-wise
 * scaling commutes with it), then invRdpcmNxN (:1737-1792): running sums along rows (RDPCM_HOR 1) or columns (RDPCM_VER 2),
 * carried in Pel = 16 bits like HM's residual buffer. */
void hmo_residual_rotate_rdpcm(int16_t* resid, int stride, int n, int rotate, int rdpcm)
{
  int x, y;
  if (rotate)
    for (y = 0; y < n / 2; y++)
      for (x = 0; x < n; x++)
      {
        int16_t* a = &resid[y * stride + x];
        int16_t* b = &resid[(n - 1 - y) * stride + (n - 1 - x)];
        const int16_t t = *a; *a = *b; *b = t;
      }
  if (rdpcm == 2)
    for (y = 1; y < n; y++) for (x = 0; x < n; x++) resid[y * stride + x] = (int16_t)(resid[y * stride + x] + resid[(y - 1) * stride + x]);
  else if (rdpcm == 1)
    for (y = 0; y < n; y++) for (x = 1; x < n; x++) resid[y * stride + x] = (int16_t)(resid[y * stride + x] + resid[y * stride + x - 1]);
}

/* ------------------------------------------------------------------------------------------------ picture hashes */
/* compCRC: TComPicYuvMD5.cpp:89-125 */
void hmo_plane_crc(int bit_depth, const int16_t* plane, int width, int height, int stride, uint8_t out[2])
{
  unsigned crc = 0xffff;
  int x, y, b, k;
  for (y = 0; y < height; y++)
    for (x = 0; x < width; x++)
      for (k = 0; k < (bit_depth > 8 ? 2 : 1); k++)
        for (b = 0; b < 8; b++)
        {
          const unsigned msb = (crc >> 15) & 1, bit = ((unsigned)plane[(size_t)y * stride + x] >> ((k ? 15 : 7) - b)) & 1;
          crc = (((crc << 1) + bit) & 0xffff) ^ (msb * 0x1021);
        }
  for (b = 0; b < 16; b++) { const unsigned msb = (crc >> 15) & 1; crc = ((crc << 1) & 0xffff) ^ (msb * 0x1021); }
  out[0] = (uint8_t)(crc >> 8); out[1] = (uint8_t)crc;
}

/* compChecksum: TComPicYuvMD5.cpp:139-165 */
void hmo_plane_checksum(int bit_depth, const int16_t* plane, int width, int height, int stride, uint8_t out[4])
{
  uint32_t sum = 0;
  int x, y;
  for (y = 0; y < height; y++)
    for (x = 0; x < width; x++)
    {
      const unsigned mask = ((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)) & 0xff, v = (uint16_t)plane[(size_t)y * stride + x];
      sum += (v & 0xff) ^ mask;
      if (bit_depth > 8) sum += (v >> 8) ^ mask;
    }
  out[0] = (uint8_t)(sum >> 24); out[1] = (uint8_t)(sum >> 16); out[2] = (uint8_t)(sum >> 8); out[3] = (uint8_t)sum;
}

/* ------------------------------------------------------------------------------------------------ interpolation */
/* TComInterpolationFilter::filter<N,isVertical,isFirst,isLast>: TLibCommon/TComInterpolationFilter.cpp:166-251 */
static void fir(int ntaps, const int* c, int vertical, int is_first, int is_last, int bit_depth,
                const int16_t* src, int src_stride, int16_t* dst, int dst_stride, int width, int height)
{
  const int head_room = imax(2, 14 - bit_depth);
  const int cs = vertical ? src_stride : 1;
  int shift = 6, offset, max_val;
  int row, col;
  src -= (ntaps / 2 - 1) * cs;
  if (is_last)
  {
    shift += is_first ? 0 : head_room;
    offset = 1 << (shift - 1);
    offset += is_first ? 0 : (8192 << 6);
    max_val = (1 << bit_depth) - 1;
  }
  else
  {
    shift -= is_first ? head_room : 0;
    offset = is_first ? -(8192 << shift) : 0;
    max_val = 0;
  }
  for (row = 0; row < height; row++)
  {
    for (col = 0; col < width; col++)
    {
      int sum;
      int16_t val;
      sum = src[col] * c[0] + src[col + cs] * c[1] + src[col + 2 * cs] * c[2] + src[col + 3 * cs] * c[3];
      if (ntaps == 8) sum += src[col + 4 * cs] * c[4] + src[col + 5 * cs] * c[5] + src[col + 6 * cs] * c[6] + src[col + 7 * cs] * c[7];
      val = (int16_t)((sum + offset) >> shift);                /* "Pel val" truncation, :239 */
      if (is_last) { if (val < 0) val = 0; if (val > max_val) val = (int16_t)max_val; }
      dst[col] = val;
    }
    src += src_stride; dst += dst_stride;
  }
}

/* filterCopy: TComInterpolationFilter.cpp:94-148 (isFirst always true on the decoder's call paths) */
static void fir_copy(int bit_depth, const int16_t* src, int src_stride, int16_t* dst, int dst_stride, int width, int height, int is_last)
{
  const int shift = imax(2, 14 - bit_depth);
  int row, col;
  for (row = 0; row < height; row++)
    for (col = 0; col < width; col++)
    {
      const int v = src[row * src_stride + col];
      dst[row * dst_stride + col] = is_last ? (int16_t)v : (int16_t)((int16_t)(v << shift) - (int16_t)8192);
    }
}

/* TComPrediction::xPredInterBlk: TLibCommon/TComPrediction.cpp:660-698.  HM reads a border-extended picture
 * (TComPicYuv::extendPicBorder, TComPicYuv.cpp:173); replication == coordinate clamp, so a block whose filter window
 * leaves the picture is first gathered into a small clamped window. */
/* csx / csy: the component's subsampling (getComponentScaleX / Y): the vector is in quarter LUMA samples, i.e. in units of
 * 1 / (4 << cs) component samples, and the chroma filter is indexed in eighths (TComPrediction.cpp:664-674, TComInterpolationFilter.cpp:344-346) */
void hmo_pred_inter_blk_fmt(int is_chroma, int csx, int csy, int bit_depth, const int16_t* ref, int ref_stride, int ref_w, int ref_h,
                            int bx, int by, int w, int h, int mvx, int mvy, int bi, int16_t* dst, int dst_stride)
{
  const int shx = 2 + csx, shy = 2 + csy;
  const int ntaps = is_chroma ? 4 : 8;
  const int xfrac = mvx & ((1 << shx) - 1), yfrac = mvy & ((1 << shy) - 1);
  const int* cx = is_chroma ? k_chroma_filter[xfrac << (1 - csx)] : k_luma_filter[xfrac];
  const int* cy = is_chroma ? k_chroma_filter[yfrac << (1 - csy)] : k_luma_filter[yfrac];
  const int ix = bx + (mvx >> shx), iy = by + (mvy >> shy);
  const int before = ntaps / 2 - 1, after = ntaps / 2;
  int16_t win[(64 + 7) * (64 + 8)];
  const int16_t* src; int ss;
  if (ix - before >= 0 && iy - before >= 0 && ix + w + after <= ref_w && iy + h + after <= ref_h)
  { src = ref + (size_t)iy * ref_stride + ix; ss = ref_stride; }
  else
  {
    int x, y;
    ss = w + ntaps - 1;
    for (y = 0; y < h + ntaps - 1; y++)
      for (x = 0; x < ss; x++)
        win[y * ss + x] = ref[(size_t)CLIP3(0, ref_h - 1, iy - before + y) * ref_stride + CLIP3(0, ref_w - 1, ix - before + x)];
    src = win + before * ss + before;
  }
  if (yfrac == 0)
  {
    if (xfrac == 0) fir_copy(bit_depth, src, ss, dst, dst_stride, w, h, !bi);
    else fir(ntaps, cx, 0, 1, !bi, bit_depth, src, ss, dst, dst_stride, w, h);
  }
  else if (xfrac == 0)
    fir(ntaps, cy, 1, 1, !bi, bit_depth, src, ss, dst, dst_stride, w, h);
  else
  {
    int16_t tmp[64 * (64 + 7)];
    fir(ntaps, cx, 0, 1, 0, bit_depth, src - before * ss, ss, tmp, w, w, h + ntaps - 1);
    fir(ntaps, cy, 1, 0, !bi, bit_depth, tmp + before * w, w, dst, dst_stride, w, h);
  }
}

void hmo_pred_inter_blk(int is_chroma, int bit_depth, const int16_t* ref, int ref_stride, int ref_w, int ref_h,
                        int bx, int by, int w, int h, int mvx, int mvy, int bi, int16_t* dst, int dst_stride)
{
  hmo_pred_inter_blk_fmt(is_chroma, is_chroma ? 1 : 0, is_chroma ? 1 : 0, bit_depth, ref, ref_stride, ref_w, ref_h, bx, by, w, h, mvx, mvy, bi, dst, dst_stride);
}

/* TComYuv::addAvg: TLibCommon/TComYuv.cpp:336-391 */
void hmo_add_avg(const int16_t* s0, const int16_t* s1, int16_t* dst, int w, int h, int stride, int bit_depth)
{
  const int shift = imax(2, 14 - bit_depth) + 1;
  const int offset = (1 << (shift - 1)) + 2 * 8192;
  const int maxv = (1 << bit_depth) - 1;
  int x, y;
  for (y = 0; y < h; y++)
    for (x = 0; x < w; x++)
    {
      int v = (s0[y * stride + x] + s1[y * stride + x] + offset) >> shift;
      dst[y * stride + x] = (int16_t)CLIP3(0, maxv, v);
    }
}

/* ------------------------------------------------------------------------------------------------ geometry */
typedef struct
{
  const hmgpu_seq_params* seq;
  const hmgpu_slice_params* slices;
  const hmgpu_ctu_meta* m;
  const hmgpu_pic_params* pp;
  int ctu, pw, parts, ctus_w, ctus_h, num_ctus, max_cu_depth;
  int w[3], h[3], bd[3];
  int fmt, csx[3], csy[3];          /* chroma_format_idc (0 handled as 4:2:0 geometry); getComponentScaleX / Y (TComChromaFormat.h:59-62) */
} geom;

static int zx(int z) { int x = 0, b; for (b = 0; b < 8; b++) x |= ((z >> (2 * b)) & 1) << b; return x; }        /* g_auiZscanToRaster column */
static int zy(int z) { int y = 0, b; for (b = 0; b < 8; b++) y |= ((z >> (2 * b + 1)) & 1) << b; return y; }
static int xy2z(int x, int y) { int z = 0, b; for (b = 0; b < 8; b++) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1)); return z; }

static void geom_init(geom* g, const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* m,
                      const hmgpu_pic_params* pp)
{
  g->seq = seq; g->slices = slices; g->m = m; g->pp = pp;
  g->ctu = 1 << seq->log2_ctu_size; g->pw = g->ctu / 4; g->parts = g->pw * g->pw;
  g->ctus_w = (seq->width + g->ctu - 1) / g->ctu; g->ctus_h = (seq->height + g->ctu - 1) / g->ctu;
  g->num_ctus = g->ctus_w * g->ctus_h;
  g->max_cu_depth = seq->log2_ctu_size - 3;           /* g_uiMaxCUDepth - g_uiAddCUDepth: CUs down to 8x8 */
  g->fmt = seq->chroma_format == 0 ? 1 : seq->chroma_format;
  g->csx[0] = g->csy[0] = 0;
  g->csx[1] = g->csx[2] = g->fmt == 3 ? 0 : 1;
  g->csy[1] = g->csy[2] = g->fmt == 1 ? 1 : 0;
  g->w[0] = seq->width; g->h[0] = seq->height;
  g->w[1] = g->w[2] = seq->width >> g->csx[1]; g->h[1] = g->h[2] = seq->height >> g->csy[1];
  g->bd[0] = seq->bit_depth_luma; g->bd[1] = g->bd[2] = seq->bit_depth_chroma;
}
/* elements of a CTU in the level / PCM arrays of component comp, and where the block of partition z starts in them (TComTU.cpp:64-76) */
static size_t ctu_elems(const geom* g, int comp) { return (size_t)(g->ctu * g->ctu) >> (g->csx[comp] + g->csy[comp]); }
static int part_elem_off(const geom* g, int comp, int z) { return (16 * z) >> (g->csx[comp] + g->csy[comp]); }
static const hmgpu_slice_params* slice_of(const geom* g, int ctu) { return &g->slices[g->m->slice_idx ? g->m->slice_idx[ctu] : 0]; }
static int slice_id(const geom* g, int ctu) { return g->m->slice_idx ? g->m->slice_idx[ctu] : 0; }
static int tile_id(const geom* g, int ctu) { return g->m->tile_idx ? g->m->tile_idx[ctu] : 0; }
#define PM(field, ctu, z) (g->m->field[(size_t)(ctu) * g->parts + (z)])

/* partitions the loop filters must leave alone: lossless CUs and, with pcm_loop_filter_disabled_flag, PCM CUs
 * (TComLoopFilter.cpp:558,629-634; TComSampleAdaptiveOffset.cpp:790) */
static int no_filter(const geom* g, int ctu, int z)
{
  const size_t i = (size_t)ctu * g->parts + z;
  if (g->m->transquant_bypass && g->m->transquant_bypass[i]) return 1;
  return g->seq->pcm_loop_filter_disable && g->m->ipcm && g->m->ipcm[i];
}

/* ------------------------------------------------------------------------------------------------ CU reconstruction */
typedef struct
{
  const geom* g;
  const hmgpu_coeffs* co;
  hmo_picture* cur;
  const hmo_picture* refs; int num_refs;
  int ctu_addr, cu_z, cu_x, cu_y, cu_size;        /* CU origin in luma samples, z index of its first partition */
  int16_t pred[3][64 * 64];
  int16_t resi[3][64 * 64];
  int16_t bi_tmp[2][3][64 * 64];                   /* the two 14-bit predictions of a bi-predicted PU (m_acYuvPred[2]: TComPrediction.h) */
} cu_ctx;

/* one leaf TU: invRecurTransformNxN leaf branch (TComTrQuant.cpp:1566-1591) */
static void tu_leaf(cu_ctx* c, int comp, int z_tu, int log2_size, int x_rel, int y_rel, int coef_off)
{
  const geom* g = c->g;
  const hmgpu_slice_params* sl = slice_of(g, c->ctu_addr);
  const int stride = c->cu_size >> g->csx[comp];
  int per, rem, flags = 0;
  const int cqo = comp == 1 ? sl->cb_qp_offset : (comp == 2 ? sl->cr_qp_offset : 0);
  const int16_t* lev = c->co->level[comp] + (size_t)c->ctu_addr * ctu_elems(g, comp) + coef_off;
  hmo_qp_param_fmt(PM(qp, c->ctu_addr, c->cu_z), comp, g->bd[comp], cqo, g->fmt, &per, &rem);       /* QpParam(cu, compID): getQP(0) */
  if (comp == 0 && PM(pred_mode, c->ctu_addr, z_tu) == HMGPU_MODE_INTRA) flags |= 1;    /* TComTU::useDST, TComTU.cpp:218 */
  const int tsb = g->m->transform_skip[comp] ? g->m->transform_skip[comp][(size_t)c->ctu_addr * g->parts + z_tu] : 0;
  /* inter blocks: the explicit mode parsed with the block (bits 1-2), honoured only while the SPS enables it (isRDPCMEnabled) */
  const int rdpcm = (g->seq->range_ext_flags & HMGPU_REXT_EXPLICIT_RDPCM) ? (tsb >> 1) & 3 : 0;
  if (tsb & 1) flags |= 2;
  if (g->m->transquant_bypass && g->m->transquant_bypass[(size_t)c->ctu_addr * g->parts + z_tu])
  {
    /* cu_transquant_bypass: the residual IS the level block (invTransformNxN, TComTrQuant.cpp:1440-1470); rotation is intra only */
    const int n = 1 << log2_size;
    int x, y;
    for (y = 0; y < n; y++) for (x = 0; x < n; x++) c->resi[comp][(y_rel + y) * stride + x_rel + x] = lev[y * n + x];
    hmo_residual_rotate_rdpcm(&c->resi[comp][y_rel * stride + x_rel], stride, n, 0, rdpcm);
    return;
  }
  /* getScalingListType (TComTrQuant.h): 3 * inter + component */
  hmo_inverse_transform_tu_sl(lev, &c->resi[comp][y_rel * stride + x_rel], stride, log2_size, g->bd[comp], per, rem, flags, sl->scaling_lists,
                              (PM(pred_mode, c->ctu_addr, z_tu) == HMGPU_MODE_INTRA ? 0 : 3) + comp);
  if (flags & 2) hmo_residual_rotate_rdpcm(&c->resi[comp][y_rel * stride + x_rel], stride, 1 << log2_size, 0, rdpcm);
}

/* the chroma block that belongs to the luma node (z, 2^log2_luma) at (xl, yl) of the CU: one square in 4:2:0 / 4:4:4, in 4:2:2 two squares
 * one above the other (invTransformNxN's TComTU::VERTICAL_SPLIT, TComTrQuant.cpp:1436-1462: both are transformed -- an uncoded one holds
 * zero levels --, the lower one with the flags of the lower half of the node's partitions), then cross-component prediction from the luma
 * residual of the same area (invRecurTransformNxN :1591-1606, crossComponentPrediction :3294-3335) */
static void tu_chroma(cu_ctx* c, int comp, int z, int log2_luma, int xl, int yl, int coded, int luma_coded)
{
  const geom* g = c->g;
  const int coff = part_elem_off(g, comp, z);
  if (coded)
  {
    if (g->fmt == 2)
    {
      const int l2 = log2_luma - 1, n = 1 << l2, zb = z + ((1 << (2 * (log2_luma - 2))) >> 1);
      tu_leaf(c, comp, z, l2, xl >> 1, yl, coff);
      tu_leaf(c, comp, zb, l2, xl >> 1, yl + n, coff + n * n);
    }
    else tu_leaf(c, comp, z, log2_luma - g->csx[comp], xl >> g->csx[comp], yl >> g->csy[comp], coff);
  }
  if (g->m->ccp_alpha[comp - 1] && luma_coded)
  {
    const int alpha = g->m->ccp_alpha[comp - 1][(size_t)c->ctu_addr * g->parts + z];
    if (alpha)
    {
      const int n = 1 << log2_luma, st = c->cu_size, diff = g->bd[0] - g->bd[comp];
      int x, y;
      for (y = 0; y < n; y++)
        for (x = 0; x < n; x++)
        {
          const int l = c->resi[0][(yl + y) * st + xl + x];
          int16_t* r = &c->resi[comp][(yl + y) * st + xl + x];
          *r = (int16_t)(*r + ((alpha * (diff >= 0 ? l >> diff : l << -diff)) >> 3));
        }
    }
  }
}

/* invRecurTransformNxN (TComTrQuant.cpp:1550-1615) with the TComTU child rules of the chroma format (TComTU.cpp:89-171) */
static void tu_recurse(cu_ctx* c, int comp, int z, int tr_depth, int log2_luma, int xl, int yl)
{
  const geom* g = c->g;
  const int a = c->ctu_addr;
  const uint8_t cbf = g->m->cbf[comp][(size_t)a * g->parts + z];
  const int ccp = comp != 0 && g->m->ccp_alpha[comp - 1] != NULL;         /* getUseCrossComponentPrediction */
  if (((cbf >> tr_depth) & 1) == 0 && !ccp) return;                      /* :1558-1564 */
  if (tr_depth == PM(tr_idx, a, z))
  {
    if (comp == 0) tu_leaf(c, 0, z, log2_luma, xl, yl, 16 * z);
    else tu_chroma(c, comp, z, log2_luma, xl, yl, (cbf >> tr_depth) & 1, (g->m->cbf[0][(size_t)a * g->parts + z] >> tr_depth) & 1);
    return;
  }
  if (comp != 0 && log2_luma == 3 && g->csx[comp])
  {
    /* the four 4x4 luma children share ONE chroma block (4x4, in 4:2:2 4x8), carried by the first child on the reconstruction path
     * (bProcessLastOfLevel == false: TComTU.cpp:141-151,171; TComTrQuant.cpp:1608); cbf tested at the child depth */
    tu_chroma(c, comp, z, 3, xl, yl, (cbf >> (tr_depth + 1)) & 1, 0);
    return;
  }
  {
    const int half = 1 << (log2_luma - 1);
    const int q = 1 << (2 * (log2_luma - 1 - 2));         /* partitions per child */
    int i;
    for (i = 0; i < 4; i++)
      tu_recurse(c, comp, z + i * q, tr_depth + 1, log2_luma - 1, xl + (i & 1) * half, yl + (i >> 1) * half);
  }
}

/* TComDataCU::clipMv: TLibCommon/TComDataCU.cpp:3102-3114 */
static void clip_mv(const cu_ctx* c, int* mvx, int* mvy)
{
  const geom* g = c->g;
  const int off = 8;
  const int hor_max = (g->seq->width + off - c->cu_x - 1) << 2, hor_min = (-g->ctu - off - c->cu_x + 1) * 4;
  const int ver_max = (g->seq->height + off - c->cu_y - 1) << 2, ver_min = (-g->ctu - off - c->cu_y + 1) * 4;
  *mvx = imin(hor_max, imax(hor_min, *mvx));
  *mvy = imin(ver_max, imax(ver_min, *mvy));
}

/* xPredInterUni for all components: TComPrediction.cpp:586-594 */
static void pred_uni(cu_ctx* c, int list, int z_pu, int xr, int yr, int w, int h, int bi, int16_t* dst[3])
{
  const geom* g = c->g;
  const hmgpu_slice_params* sl = slice_of(g, c->ctu_addr);
  const size_t pi = (size_t)c->ctu_addr * g->parts + z_pu;
  const int ref_idx = g->m->ref_idx[list][pi];
  int mvx = g->m->mv[list][pi * 2], mvy = g->m->mv[list][pi * 2 + 1];
  const hmo_picture* rp = &c->refs[sl->ref_pic[list][ref_idx]];
  int comp;
  clip_mv(c, &mvx, &mvy);
  for (comp = 0; comp < 3; comp++)
  {
    const int sx = g->csx[comp], sy = g->csy[comp];
    const int stride = c->cu_size >> sx;
    hmo_pred_inter_blk_fmt(comp != 0, sx, sy, g->bd[comp], rp->plane[comp], g->w[comp], g->w[comp], g->h[comp],
                           (c->cu_x + xr) >> sx, (c->cu_y + yr) >> sy, w >> sx, h >> sy, mvx, mvy, bi,
                           dst[comp] + (yr >> sy) * stride + (xr >> sx), stride);
  }
}

/* motionCompensation for one PU: TComPrediction.cpp:514-584 (REF_PIC_LIST_X path, no weighted prediction) */
static void pred_pu(cu_ctx* c, int z_pu, int xr, int yr, int w, int h)
{
  const geom* g = c->g;
  const hmgpu_slice_params* sl = slice_of(g, c->ctu_addr);
  const size_t pi = (size_t)c->ctu_addr * g->parts + z_pu;
  const int r0 = g->m->ref_idx[0][pi], r1 = g->m->ref_idx[1][pi];
  int16_t* out[3] = { c->pred[0], c->pred[1], c->pred[2] };
  int identical = 0;
  if (sl->weighted_pred)
  {
    /* explicit weighted prediction: xPredInterBi with bi = true for every used list, then xWeightedPredictionBi / Uni
     * (TComPrediction.cpp:596-644; TComWeightPrediction.cpp:44-57 weightBidir / weightUnidir, :211-271 getWpScaling).
     * xCheckIdenticalMotion is off for B slices with weighted_bipred_flag (:499). */
    int16_t* a[3] = { c->bi_tmp[0][0], c->bi_tmp[0][1], c->bi_tmp[0][2] };
    int16_t* b[3] = { c->bi_tmp[1][0], c->bi_tmp[1][1], c->bi_tmp[1][2] };
    int comp, x, y;
    if (r0 >= 0) pred_uni(c, 0, z_pu, xr, yr, w, h, 1, a);
    if (r1 >= 0) pred_uni(c, 1, z_pu, xr, yr, w, h, 1, b);
    for (comp = 0; comp < 3; comp++)
    {
      const int sx = g->csx[comp], sy = g->csy[comp], stride = c->cu_size >> sx;
      const int o = (yr >> sy) * stride + (xr >> sx);
      const int bd = g->bd[comp], maxv = (1 << bd) - 1;
      const int shift_num = imax(2, 14 - bd), log2wd = sl->wp_log2_denom[comp ? 1 : 0];
      for (y = 0; y < (h >> sy); y++)
        for (x = 0; x < (w >> sx); x++)
        {
          const int i = o + y * stride + x;
          int v;
          if (r0 >= 0 && r1 >= 0)
          {
            const int shift = log2wd + 1 + shift_num, round = 1 << (shift - 1);
            const int off = sl->wp_offset[0][r0][comp] + sl->wp_offset[1][r1][comp];
            v = (sl->wp_weight[0][r0][comp] * (a[comp][i] + 8192) + sl->wp_weight[1][r1][comp] * (b[comp][i] + 8192) + round +
                 (off << (shift - 1))) >> shift;
          }
          else
          {
            const int l = r0 >= 0 ? 0 : 1, r = l ? r1 : r0;
            const int shift = log2wd + shift_num, round = shift > 0 ? 1 << (shift - 1) : 0;
            const int p = l ? b[comp][i] : a[comp][i];
            v = ((sl->wp_weight[l][r][comp] * (p + 8192) + round) >> shift) + sl->wp_offset[l][r][comp];
          }
          out[comp][i] = (int16_t)CLIP3(0, maxv, v);
        }
    }
    return;
  }
  if (sl->slice_type == HMGPU_B_SLICE && r0 >= 0 && r1 >= 0)                      /* xCheckIdenticalMotion :497-512 */
    identical = sl->ref_poc[0][r0] == sl->ref_poc[1][r1] && g->m->mv[0][pi * 2] == g->m->mv[1][pi * 2] &&
                g->m->mv[0][pi * 2 + 1] == g->m->mv[1][pi * 2 + 1];
  if (identical) { pred_uni(c, 0, z_pu, xr, yr, w, h, 0, out); return; }
  if (r0 >= 0 && r1 >= 0)
  {
    /* xPredInterBi + xWeightedAverage -> addAvg: :596-644, :700-714 */
    int16_t* a[3] = { c->bi_tmp[0][0], c->bi_tmp[0][1], c->bi_tmp[0][2] };
    int16_t* b[3] = { c->bi_tmp[1][0], c->bi_tmp[1][1], c->bi_tmp[1][2] };
    int comp;
    pred_uni(c, 0, z_pu, xr, yr, w, h, 1, a);
    pred_uni(c, 1, z_pu, xr, yr, w, h, 1, b);
    for (comp = 0; comp < 3; comp++)
    {
      const int sx = g->csx[comp], sy = g->csy[comp], stride = c->cu_size >> sx;
      const int o = (yr >> sy) * stride + (xr >> sx);
      hmo_add_avg(a[comp] + o, b[comp] + o, out[comp] + o, w >> sx, h >> sy, stride, g->bd[comp]);
    }
    return;
  }
  pred_uni(c, r0 >= 0 ? 0 : 1, z_pu, xr, yr, w, h, 0, out);
}

/* TComDataCU::getPartIndexAndSize: TLibCommon/TComDataCU.cpp:2178-2216.  returns number of PUs */
static int pu_layout(int part_size, int cu_size, int num_part, int z_off[4], int xr[4], int yr[4], int w[4], int h[4])
{
  const int s = cu_size, hs = s >> 1, q = s >> 2;
  int n = 1, i;
  for (i = 0; i < 4; i++) { z_off[i] = 0; xr[i] = yr[i] = 0; w[i] = s; h[i] = s; }
  switch (part_size)
  {
    case HMGPU_SIZE_2NxN:  n = 2; h[0] = h[1] = hs; yr[1] = hs; z_off[1] = num_part >> 1; break;
    case HMGPU_SIZE_Nx2N:  n = 2; w[0] = w[1] = hs; xr[1] = hs; z_off[1] = num_part >> 2; break;
    case HMGPU_SIZE_NxN:   n = 4; for (i = 0; i < 4; i++) { w[i] = h[i] = hs; xr[i] = (i & 1) * hs; yr[i] = (i >> 1) * hs; z_off[i] = (num_part >> 2) * i; } break;
    case HMGPU_SIZE_2NxnU: n = 2; h[0] = q; h[1] = q + hs; yr[1] = q; z_off[1] = num_part >> 3; break;
    case HMGPU_SIZE_2NxnD: n = 2; h[0] = q + hs; h[1] = q; yr[1] = q + hs; z_off[1] = (num_part >> 1) + (num_part >> 3); break;
    case HMGPU_SIZE_nLx2N: n = 2; w[0] = q; w[1] = q + hs; xr[1] = q; z_off[1] = num_part >> 4; break;
    case HMGPU_SIZE_nRx2N: n = 2; w[0] = q + hs; w[1] = q; xr[1] = q + hs; z_off[1] = (num_part >> 2) + (num_part >> 4); break;
    default: break;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------------ intra prediction
 * TDecCu::xReconIntraQT / xIntraRecQT / xIntraRecBlk (TDecCu.cpp:484-730), TComPrediction::initAdiPatternChType and
 * fillReferenceSamples (TComPattern.cpp:107-500), predIntraAng / xPredIntraAng / xPredIntraPlanar / xDCPredFiltering
 * (TComPrediction.cpp:182-228, 245-491, 746-840).  4:2:0, no RDPCM / cross-component prediction / PCM. */

/* is the 4x4 luma partition at luma sample (px, py) usable as intra reference for the TU whose first partition is
 * (cur_ctu, cur_z)?  TComDataCU::getPULeft/Above/AboveLeft/AboveRightAdi/BelowLeftAdi (TComDataCU.cpp:1177-1530) boil
 * down to: inside the picture, decoded before the current TU, same slice and tile; with constrained intra prediction
 * also "is intra" (TComPattern.cpp:558-700) */
static int intra_avail(const geom* g, int cur_ctu, int cur_z, int px, int py, int cip)
{
  int nctu, nz;
  if (px < 0 || py < 0 || px >= g->seq->width || py >= g->seq->height) return 0;
  nctu = (py >> g->seq->log2_ctu_size) * g->ctus_w + (px >> g->seq->log2_ctu_size);
  nz = xy2z((px & (g->ctu - 1)) >> 2, (py & (g->ctu - 1)) >> 2);
  if (nctu == cur_ctu) { if (nz >= cur_z) return 0; }
  else
  {
    if (nctu > cur_ctu) return 0;
    if (slice_id(g, nctu) != slice_id(g, cur_ctu) || tile_id(g, nctu) != tile_id(g, cur_ctu)) return 0;
  }
  if (cip && PM(pred_mode, nctu, nz) != HMGPU_MODE_INTRA) return 0;
  return 1;
}

/* reference line of a TU, 4N+1 samples: [0, 2N) left column from the bottom-most below-left sample up to the row of the
 * TU's first line, [2N] the corner, (2N, 4N] the row above from left to right (fillReferenceSamples with the
 * substitution of 8.4.4.2.2, which HM performs unit by unit: TComPattern.cpp:336-478) */
static void intra_ref_line(const cu_ctx* c, int comp, int z_tu, int n, int x0, int y0, int* line)
{
  const geom* g = c->g;
  const int sx = g->csx[comp], sy = g->csy[comp];
  const int ux = 4 >> sx, uy = 4 >> sy;                                /* samples of the component per 4x4 luma partition, along x / y */
  const int cip = slice_of(g, c->ctu_addr)->constrained_intra_pred;
  const int lx = x0 << sx, ly = y0 << sy;                              /* TU origin in luma samples */
  const int16_t* pl = c->cur->plane[comp];
  const int stride = g->w[comp];
  const int total = 4 * n + 1, corner = 2 * n;
  uint8_t ok[4 * 32 + 1];
  int i, any = 0;
  for (i = 0; i < 2 * n / uy; i++)
  {
    const int al = intra_avail(g, c->ctu_addr, z_tu, lx - 4, ly + 4 * i, cip);        /* left column, unit i from the top */
    int k;
    for (k = 0; k < uy; k++) ok[corner - 1 - (i * uy + k)] = (uint8_t)al;
    any |= al;
  }
  for (i = 0; i < 2 * n / ux; i++)
  {
    const int aa = intra_avail(g, c->ctu_addr, z_tu, lx + 4 * i, ly - 4, cip);        /* row above, unit i from the left */
    int k;
    for (k = 0; k < ux; k++) ok[corner + 1 + i * ux + k] = (uint8_t)aa;
    any |= aa;
  }
  ok[corner] = (uint8_t)intra_avail(g, c->ctu_addr, z_tu, lx - 4, ly - 4, cip);
  any |= ok[corner];
  if (!any) { for (i = 0; i < total; i++) line[i] = 1 << (g->bd[comp] - 1); return; }
  for (i = 0; i < total; i++)
  {
    if (!ok[i]) continue;
    if (i < corner) { const int r = corner - 1 - i; line[i] = pl[(size_t)(y0 + r) * stride + x0 - 1]; }
    else if (i == corner) line[i] = pl[(size_t)(y0 - 1) * stride + x0 - 1];
    else line[i] = pl[(size_t)(y0 - 1) * stride + x0 + (i - corner - 1)];
  }
  if (!ok[0])
  {
    int first = 1;
    while (first < total && !ok[first]) first++;
    line[0] = line[first];
  }
  for (i = 1; i < total; i++) if (!ok[i]) line[i] = line[i - 1];
}

/* TComPrediction::filteringIntraReferenceSamples + the smoothing of initAdiPatternChType (TComPattern.cpp:186-296, 531-556) */
static void intra_smooth(const geom* g, int comp, int mode, int n, int log2n, const int* in, int* out)
{
  static const int thr[6] = { 0, 0, 10, 7, 1, 0 };                     /* m_aucIntraFilter by log2 size: TComPrediction.cpp:49-66 */
  const int total = 4 * n + 1, corner = 2 * n;
  int i, filt = 0;
  /* filterIntraReferenceSamples (TComChromaFormat.h:150-153): luma, and chroma where it is not subsampled (4:4:4) */
  if ((comp == 0 || g->fmt == 3) && mode != 1 /* DC_IDX */ && !(g->seq->range_ext_flags & HMGPU_REXT_INTRA_SMOOTHING_DISABLED))
  {
    const int d0 = iabs(mode - 10), d1 = iabs(mode - 26);
    filt = imin(d0, d1) > thr[log2n];
  }
  for (i = 0; i < total; i++) out[i] = in[i];
  if (!filt) return;
  if (comp == 0 && g->seq->strong_intra_smoothing && n == 32)            /* isLuma(chType) && getUseStrongIntraSmoothing: TComPattern.cpp:196 */
  {
    const int t = 1 << (g->bd[0] - 5);
    const int bl = in[0], tl = in[corner], tr = in[total - 1];
    if (iabs(bl + tl - 2 * in[n]) < t && iabs(tl + tr - 2 * in[corner + n]) < t)
    {
      for (i = 1; i < 2 * n; i++)
      {
        out[i] = ((2 * n - i) * bl + i * tl + n) >> (log2n + 1);
        out[corner + i] = ((2 * n - i) * tl + i * tr + n) >> (log2n + 1);
      }
      return;
    }
  }
  for (i = 1; i < total - 1; i++) out[i] = (in[i - 1] + 2 * in[i] + in[i + 1] + 2) >> 2;
}

/* predIntraAng: planar, DC (+ edge filter) and the 33 angular modes; pred is n x n, stride n */
static void intra_predict(int comp, int bd, int mode, int n, int log2n, const int* line, int16_t* pred, int edge_filters)
{
  const int corner = 2 * n;
  const int* left = line + corner - 1;          /* left[-r] = row r */
  const int* above = line + corner + 1;         /* above[x] */
  const int maxv = (1 << bd) - 1;
  const int edge = comp == 0 && n <= 16;        /* MAXIMUM_INTRA_FILTERED_WIDTH: TypeDef.h:117 */
  int x, y;
  if (mode == 0)
  {
    const int bl = left[-n], tr = above[n];
    for (y = 0; y < n; y++)
      for (x = 0; x < n; x++)
      {
        const int hor = (left[-y] << log2n) + n + (x + 1) * (tr - left[-y]);
        const int ver = (above[x] << log2n) + (y + 1) * (bl - above[x]);
        pred[y * n + x] = (int16_t)((hor + ver) >> (log2n + 1));
      }
    return;
  }
  if (mode == 1)
  {
    int sum = n, dc;
    for (x = 0; x < n; x++) sum += above[x] + left[-x];
    dc = sum >> (log2n + 1);
    for (y = 0; y < n * n; y++) pred[y] = (int16_t)dc;
    if (edge)
    {
      pred[0] = (int16_t)((above[0] + left[0] + 2 * dc + 2) >> 2);
      for (x = 1; x < n; x++) pred[x] = (int16_t)((above[x] + 3 * dc + 2) >> 2);
      for (y = 1; y < n; y++) pred[y * n] = (int16_t)((left[-y] + 3 * dc + 2) >> 2);
    }
    return;
  }
  {
    static const int ang_table[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    static const int inv_table[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
    const int ver = mode >= 18;
    const int am = ver ? mode - 26 : -(mode - 10);
    const int ang = (am < 0 ? -1 : 1) * ang_table[iabs(am)], inv = inv_table[iabs(am)];
    int ref_buf[3 * 32 + 2];
    int* ref = ref_buf + 32;                     /* ref[0] = corner, ref[1..] main side, ref[-1..] projected side samples */
    int k;
    /* main = above for vertical modes, left for horizontal ones; side = the other */
#define MAIN(i) (ver ? line[corner + (i)] : line[corner - (i)])
#define SIDE(i) (ver ? line[corner - (i)] : line[corner + (i)])
    if (ang < 0)
    {
      int inv_sum = 128;
      for (k = 0; k <= n; k++) ref[k] = MAIN(k);
      for (k = -1; k > ((n * ang) >> 5); k--) { inv_sum += inv; ref[k] = SIDE(inv_sum >> 8); }
    }
    else
      for (k = 0; k <= 2 * n; k++) ref[k] = MAIN(k);
    for (y = 0; y < n; y++)
    {
      const int pos = (y + 1) * ang, di = pos >> 5, df = pos & 31;
      for (x = 0; x < n; x++)
      {
        int v = df ? ((32 - df) * ref[x + di + 1] + df * ref[x + di + 2] + 16) >> 5 : ref[x + di + 1];
        if (ang == 0 && edge && edge_filters && x == 0) v = CLIP3(0, maxv, v + ((SIDE(y + 1) - SIDE(0)) >> 1));
        if (ver) pred[y * n + x] = (int16_t)v; else pred[x * n + y] = (int16_t)v;
      }
    }
#undef MAIN
#undef SIDE
  }
}

static const unsigned char k_chroma422_mode[36] = {                       /* g_chroma422IntraAngleMappingTable: TComRom.cpp:534-536 */
  0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31, 36 };

/* xIntraRecBlk: prediction, residual, reconstruction straight into the picture (later TUs predict from it).  The block is the square of
 * 2^log2n samples of component comp whose first 4x4 luma partition is z_tu (the lower square of a 4:2:2 chroma block: the first partition of
 * the lower half); z_mode: the partition the block's mode and cross-component weight are stored at (the upper square's) */
static void intra_tu(cu_ctx* c, int comp, int z_tu, int z_mode, int log2n, int cbf_depth, int coef_off)
{
  const geom* g = c->g;
  const int a = c->ctu_addr, sx = g->csx[comp], sy = g->csy[comp], n = 1 << log2n;
  const int x0 = ((a % g->ctus_w) * g->ctu + zx(z_tu) * 4) >> sx, y0 = ((a / g->ctus_w) * g->ctu + zy(z_tu) * 4) >> sy;
  const hmgpu_slice_params* sl = slice_of(g, a);
  int mode = comp == 0 ? g->m->intra_dir[0][(size_t)a * g->parts + z_mode] : g->m->intra_dir[1][(size_t)a * g->parts + z_mode];
  int line[4 * 32 + 1], fl[4 * 32 + 1];
  int16_t pred[32 * 32], resi[32 * 32];
  int16_t* dst = c->cur->plane[comp] + (size_t)y0 * g->w[comp] + x0;
  const int maxv = (1 << g->bd[comp]) - 1;
  int x, y;
  if (comp != 0 && mode == 36) {   /* DM_CHROMA_IDX: TDecCu.cpp:523-524 with getChromasCorrespondingPULumaIdx (TComChromaFormat.h:129-132): 4:4:4: the luma
                                      mode of the same partition; else the luma mode of the first partition of the minimum-size CU -- intra NxN only
                                      exists at that size, every other CU carries one mode throughout, so "first partition of the CU" names the same mode */
    const int cu_parts = g->parts >> (2 * g->m->depth[(size_t)a * g->parts + z_mode]);
    mode = g->m->intra_dir[0][(size_t)a * g->parts + (g->fmt == 3 ? z_mode : (z_mode & ~(cu_parts - 1)))];
  }
  if (comp != 0 && g->fmt == 2) mode = k_chroma422_mode[mode];            /* uiChFinalMode: TDecCu.cpp:525 */
  intra_ref_line(c, comp, z_tu, n, x0, y0, line);
  intra_smooth(g, comp, mode, n, log2n, line, fl);
  {
    /* implicit RDPCM in a lossless CU switches the edge filters of the horizontal / vertical modes off (TComPrediction.cpp:476) */
    const int byp = g->m->transquant_bypass && g->m->transquant_bypass[(size_t)a * g->parts + z_tu];
    intra_predict(comp, g->bd[comp], mode, n, log2n, fl, pred, !(byp && (g->seq->range_ext_flags & HMGPU_REXT_IMPLICIT_RDPCM)));
  }
  memset(resi, 0, sizeof(int16_t) * n * n);
  if ((g->m->cbf[comp][(size_t)a * g->parts + z_tu] >> cbf_depth) & 1)
  {
    int per, rem, flags = comp == 0 ? 1 : 0;
    const int cqo = comp == 1 ? sl->cb_qp_offset : (comp == 2 ? sl->cr_qp_offset : 0);
    const int16_t* lev = c->co->level[comp] + (size_t)a * ctu_elems(g, comp) + coef_off;
    const int byp = g->m->transquant_bypass && g->m->transquant_bypass[(size_t)a * g->parts + z_tu];
    if (byp) memcpy(resi, lev, sizeof(int16_t) * n * n);
    else
    {
    hmo_qp_param_fmt(PM(qp, a, c->cu_z), comp, g->bd[comp], cqo, g->fmt, &per, &rem);
    if (g->m->transform_skip[comp] && (g->m->transform_skip[comp][(size_t)a * g->parts + z_tu] & 1)) flags |= 2;
    hmo_inverse_transform_tu_sl(lev, resi, n, log2n, g->bd[comp], per, rem, flags, sl->scaling_lists, comp);
    }
    if (byp || (flags & 2))
    {
      /* isNonTransformedResidualRotated (TComTU.cpp:227-233): 4x4 intra; implicit RDPCM follows the final prediction mode (invRdpcmNxN) */
      const int rx = g->seq->range_ext_flags;
      hmo_residual_rotate_rdpcm(resi, n, n, (rx & HMGPU_REXT_ROTATION) && n == 4,
                                (rx & HMGPU_REXT_IMPLICIT_RDPCM) ? (mode == 10 ? 1 : (mode == 26 ? 2 : 0)) : 0);
    }
  }
  {
    /* the luma residual of the CU is kept for the cross-component prediction of its chroma (xIntraRecBlk, TDecCu.cpp:583-612: 4:4:4 only) */
    const int rx = x0 - (c->cu_x >> sx), ry = y0 - (c->cu_y >> sy), st = c->cu_size;
    if (comp == 0)
      for (y = 0; y < n; y++) memcpy(&c->resi[0][(ry + y) * st + rx], &resi[y * n], sizeof(int16_t) * n);
    else if (g->m->ccp_alpha[comp - 1])
    {
      const int alpha = g->m->ccp_alpha[comp - 1][(size_t)a * g->parts + z_mode], diff = g->bd[0] - g->bd[comp];
      if (alpha)
        for (y = 0; y < n; y++)
          for (x = 0; x < n; x++)
          {
            const int l = c->resi[0][(ry + y) * st + rx + x];
            resi[y * n + x] = (int16_t)(resi[y * n + x] + ((alpha * (diff >= 0 ? l >> diff : l << -diff)) >> 3));
          }
    }
  }
  for (y = 0; y < n; y++)
    for (x = 0; x < n; x++)
      dst[(size_t)y * g->w[comp] + x] = (int16_t)CLIP3(0, maxv, pred[y * n + x] + resi[y * n + x]);
}

/* the chroma of the luma node (z, 2^log2_luma): Cb then Cr, in 4:2:2 each as two squares, the upper one first (xIntraRecBlk's
 * TComTU::VERTICAL_SPLIT, TDecCu.cpp:505-520: the lower square predicts from the upper one's reconstruction) */
static void intra_chroma(cu_ctx* c, int z, int log2_luma, int cbf_depth)
{
  const geom* g = c->g;
  int comp;
  for (comp = 1; comp < 3; comp++)
  {
    const int coff = part_elem_off(g, comp, z);
    if (g->fmt == 2)
    {
      const int l2 = log2_luma - 1, n = 1 << l2, zb = z + ((1 << (2 * (log2_luma - 2))) >> 1);
      intra_tu(c, comp, z, z, l2, cbf_depth, coff);
      intra_tu(c, comp, zb, z, l2, cbf_depth, coff + n * n);
    }
    else intra_tu(c, comp, z, z, log2_luma - g->csx[comp], cbf_depth, coff);
  }
}

/* xIntraRecQT for one channel type (ch 0 luma, 1 chroma = Cb then Cr per TU) */
static void intra_recurse(cu_ctx* c, int ch, int z, int tr_depth, int log2_luma)
{
  const geom* g = c->g;
  if (tr_depth == PM(tr_idx, c->ctu_addr, z))
  {
    if (ch == 0) intra_tu(c, 0, z, z, log2_luma, tr_depth, 16 * z);
    else intra_chroma(c, z, log2_luma, tr_depth);
    return;
  }
  if (ch != 0 && log2_luma == 3 && g->csx[1])
  {
    /* four 4x4 luma TUs, one chroma block per component (4x4; 4:2:2: 4x8) with the first of them (TComTU.cpp:141-171), cbf at the child depth */
    intra_chroma(c, z, 3, tr_depth + 1);
    return;
  }
  {
    const int q = 1 << (2 * (log2_luma - 1 - 2));
    int i;
    for (i = 0; i < 4; i++) intra_recurse(c, ch, z + i * q, tr_depth + 1, log2_luma - 1);
  }
}

/* TDecCu::xDecompressCU: TLibDecoder/TDecCu.cpp:373-447 */
static void decompress_cu(cu_ctx* c, int z, int depth, int64_t* n_intra)
{
  const geom* g = c->g;
  const int a = c->ctu_addr;
  const int size = g->ctu >> depth;
  const int lx = (a % g->ctus_w) * g->ctu + zx(z) * 4, ty = (a / g->ctus_w) * g->ctu + zy(z) * 4;
  const int boundary = (lx + size - 1 >= g->seq->width) || (ty + size - 1 >= g->seq->height);
  const int num_part = g->parts >> (2 * depth);
  if ((depth < PM(depth, a, z) && depth < g->max_cu_depth) || boundary)
  {
    const int q = g->parts >> (2 * (depth + 1));
    int i;
    for (i = 0; i < 4; i++)
    {
      const int zi = z + i * q;
      const int x = (a % g->ctus_w) * g->ctu + zx(zi) * 4, y = (a / g->ctus_w) * g->ctu + zy(zi) * 4;
      if (x < g->seq->width && y < g->seq->height) decompress_cu(c, zi, depth + 1, n_intra);
    }
    return;
  }
  c->cu_z = z; c->cu_x = lx; c->cu_y = ty; c->cu_size = size;
  if (g->m->ipcm && g->m->ipcm[(size_t)a * g->parts + z] && c->co->pcm_sample[0])
  {
    /* xReconPCM / xDecodePCMTexture (TDecCu.cpp:770-830): the transmitted samples, shifted up to the coding bit depth */
    int comp, x, y;
    *n_intra += num_part;
    for (comp = 0; comp < 3; comp++)
    {
      const int cw = size >> g->csx[comp], chh = size >> g->csy[comp];
      const int shift = g->bd[comp] - (comp ? g->seq->pcm_bit_depth_chroma : g->seq->pcm_bit_depth_luma);
      const int16_t* src = c->co->pcm_sample[comp] + (size_t)a * ctu_elems(g, comp) + part_elem_off(g, comp, z);
      int16_t* dst = c->cur->plane[comp] + (size_t)(ty >> g->csy[comp]) * g->w[comp] + (lx >> g->csx[comp]);
      for (y = 0; y < chh; y++) for (x = 0; x < cw; x++) dst[(size_t)y * g->w[comp] + x] = (int16_t)(src[y * cw + x] << shift);
    }
    return;
  }
  if (PM(pred_mode, a, z) == HMGPU_MODE_INTRA)
  {
    *n_intra += num_part;
    if (!g->m->intra_dir[0] || !g->m->intra_dir[1]) return;      /* no intra modes supplied: the CU's samples are left alone */
    intra_recurse(c, 0, z, 0, g->seq->log2_ctu_size - depth);       /* xReconIntraQT: luma of the whole CU, then chroma (:665-690) */
    intra_recurse(c, 1, z, 0, g->seq->log2_ctu_size - depth);
    return;
  }
  {
    /* xReconInter: TDecCu.cpp:449-482 */
    int z_off[4], xr[4], yr[4], w[4], h[4], i, comp, x, y;
    const int n = pu_layout(PM(part_size, a, z), size, num_part, z_off, xr, yr, w, h);
    for (i = 0; i < n; i++) pred_pu(c, z + z_off[i], xr[i], yr[i], w[i], h[i]);
    for (comp = 0; comp < 3; comp++)                                             /* m_ppcYuvResi->clear(): :413 */
      memset(c->resi[comp], 0, sizeof(int16_t) * (size >> g->csx[comp]) * (size >> g->csy[comp]));
    for (comp = 0; comp < 3; comp++)
    {
      const int cw = size >> g->csx[comp], chh = size >> g->csy[comp];
      tu_recurse(c, comp, z, 0, g->seq->log2_ctu_size - depth, 0, 0);          /* xDecodeInterTexture: :743-757 (luma first: chroma may predict from its residual) */
      {
        /* addClip (TComYuv.cpp:264-299) then xCopyToPic (TDecCu.cpp:734); adding an all-zero residual == the copy branch */
        const int maxv = (1 << g->bd[comp]) - 1;
        int16_t* dst = c->cur->plane[comp] + (size_t)(ty >> g->csy[comp]) * g->w[comp] + (lx >> g->csx[comp]);
        for (y = 0; y < chh; y++)
          for (x = 0; x < cw; x++)
          {
            const int v = c->pred[comp][y * cw + x] + c->resi[comp][y * cw + x];
            dst[(size_t)y * g->w[comp] + x] = (int16_t)CLIP3(0, maxv, v);
          }
      }
    }
  }
}

int hmo_decompress_ctus(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                        const hmgpu_coeffs* coeffs, hmo_picture* cur, const hmo_picture* refs, int num_refs,
                        int first_ctu, int num_ctus, int64_t* n_intra_parts)
{
  geom g;
  cu_ctx* c = (cu_ctx*)malloc(sizeof(cu_ctx));
  int a;
  int64_t n_intra = 0;
  if (!c) return HMGPU_ENOMEM;
  geom_init(&g, seq, slices, meta, NULL);
  c->g = &g; c->co = coeffs; c->cur = cur; c->refs = refs; c->num_refs = num_refs;
  for (a = first_ctu; a < first_ctu + num_ctus; a++)
  {
    c->ctu_addr = a;
    decompress_cu(c, 0, 0, &n_intra);
  }
  if (n_intra_parts) *n_intra_parts = n_intra;
  free(c);
  return HMGPU_OK;
}

/* ------------------------------------------------------------------------------------------------ deblocking */
typedef struct
{
  const geom* g;
  hmo_picture* pic;
  uint8_t bs[2][256];          /* m_aapucBS */
  uint8_t edge[2][256];        /* m_aapbEdgeFilter */
  int internal_edge, left_edge, top_edge;      /* m_stLFCUParam */
} lf_ctx;

/* neighbour partition across the left / top border: TComDataCU::getPULeft / getPUAbove (TComDataCU.cpp:1177-1259)
 * with bEnforceSliceRestriction = !lfCrossSlice of the CURRENT CTU's slice, bEnforceTileRestriction = !lfCrossTiles.
 * returns 0 if "NULL" */
static int neighbour(const geom* g, int ctu, int z, int dir /*0 left 1 above*/, int* nctu, int* nz)
{
  const int x = zx(z), y = zy(z);
  if (dir == 0 ? x > 0 : y > 0) { *nctu = ctu; *nz = dir == 0 ? xy2z(x - 1, y) : xy2z(x, y - 1); return 1; }
  {
    const int cx = ctu % g->ctus_w, cy = ctu / g->ctus_w;
    int n;
    if (dir == 0 ? cx == 0 : cy == 0) return 0;
    n = dir == 0 ? ctu - 1 : ctu - g->ctus_w;
    if (!slice_of(g, ctu)->lf_across_slices && slice_id(g, n) != slice_id(g, ctu)) return 0;
    if (g->pp && !g->pp->lf_across_tiles && tile_id(g, n) != tile_id(g, ctu)) return 0;
    *nctu = n; *nz = dir == 0 ? xy2z(g->pw - 1, y) : xy2z(x, g->pw - 1);
    return 1;
  }
}

/* xSetEdgefilterTU: TComLoopFilter.cpp:269-291 */
static void set_edge_tu(lf_ctx* l, int ctu, int z, int tr_depth, int log2_luma)
{
  const geom* g = l->g;
  if (PM(tr_idx, ctu, z) > tr_depth)
  {
    const int q = 1 << (2 * (log2_luma - 1 - 2));
    int i;
    for (i = 0; i < 4; i++) set_edge_tu(l, ctu, z + i * q, tr_depth + 1, log2_luma - 1);
    return;
  }
  {
    const int n = (1 << log2_luma) / 4, x0 = zx(z), y0 = zy(z);
    int u;
    for (u = 0; u < n; u++)
    {
      const int zv = xy2z(x0, y0 + u), zh = xy2z(x0 + u, y0);
      l->edge[0][zv] = (uint8_t)l->internal_edge; l->bs[0][zv] = (uint8_t)l->internal_edge;     /* xSetEdgefilterMultiple, iEdgeIdx 0 */
      l->edge[1][zh] = (uint8_t)l->internal_edge; l->bs[1][zh] = (uint8_t)l->internal_edge;
    }
  }
}

/* xSetEdgefilterMultiple for a CU-wide edge at offset e (partition units): TComLoopFilter.cpp:236-267 */
static void set_edge_cu(lf_ctx* l, int z_cu, int cu_parts_w, int dir, int e, int value)
{
  const int x0 = zx(z_cu), y0 = zy(z_cu);
  int u;
  for (u = 0; u < cu_parts_w; u++)
  {
    const int z = dir == 0 ? xy2z(x0 + e, y0 + u) : xy2z(x0 + u, y0 + e);
    l->edge[dir][z] = (uint8_t)value;
    if (e == 0) l->bs[dir][z] = (uint8_t)value;
  }
}

/* xGetBoundaryStrengthSingle: TComLoopFilter.cpp:411-537 */
static void boundary_strength(lf_ctx* l, int ctu, int dir, int zq)
{
  const geom* g = l->g;
  int pctu = 0, zp = 0, bs = 0;
  size_t qi, pi;
  int p_intra, q_intra;
  neighbour(g, ctu, zq, dir, &pctu, &zp);
  qi = (size_t)ctu * g->parts + zq; pi = (size_t)pctu * g->parts + zp;
  p_intra = g->m->pred_mode[pi] == HMGPU_MODE_INTRA; q_intra = g->m->pred_mode[qi] == HMGPU_MODE_INTRA;
  if (p_intra || q_intra) bs = 2;
  if (!p_intra && !q_intra)
  {
    if (l->bs[dir][zq] && (((g->m->cbf[0][qi] >> g->m->tr_idx[qi]) & 1) || ((g->m->cbf[0][pi] >> g->m->tr_idx[pi]) & 1)))
      bs = 1;
    else
    {
      const hmgpu_slice_params* sq = slice_of(g, ctu);
      const hmgpu_slice_params* sp = slice_of(g, pctu);
      int mvp[2][2], mvq[2][2], rp[2], rq[2], k;
      for (k = 0; k < 2; k++)
      {
        const int ip = g->m->ref_idx[k][pi], iq = g->m->ref_idx[k][qi];
        rp[k] = ip < 0 ? HMGPU_NO_PIC : sp->ref_pic[k][ip];
        rq[k] = iq < 0 ? HMGPU_NO_PIC : sq->ref_pic[k][iq];
        mvp[k][0] = ip < 0 ? 0 : g->m->mv[k][pi * 2]; mvp[k][1] = ip < 0 ? 0 : g->m->mv[k][pi * 2 + 1];
        mvq[k][0] = iq < 0 ? 0 : g->m->mv[k][qi * 2]; mvq[k][1] = iq < 0 ? 0 : g->m->mv[k][qi * 2 + 1];
      }
      if (sq->slice_type == HMGPU_B_SLICE || sp->slice_type == HMGPU_B_SLICE)
      {
#define MVD(a, b) (iabs((a)[0] - (b)[0]) >= 4 || iabs((a)[1] - (b)[1]) >= 4)
        if ((rp[0] == rq[0] && rp[1] == rq[1]) || (rp[0] == rq[1] && rp[1] == rq[0]))
        {
          if (rp[0] != rp[1])
          {
            if (rp[0] == rq[0]) bs = (MVD(mvq[0], mvp[0]) || MVD(mvq[1], mvp[1])) ? 1 : 0;
            else bs = (MVD(mvq[1], mvp[0]) || MVD(mvq[0], mvp[1])) ? 1 : 0;
          }
          else
            bs = ((MVD(mvq[0], mvp[0]) || MVD(mvq[1], mvp[1])) && (MVD(mvq[1], mvp[0]) || MVD(mvq[0], mvp[1]))) ? 1 : 0;
        }
        else bs = 1;
      }
      else
        bs = (rp[0] != rq[0] || MVD(mvq[0], mvp[0])) ? 1 : 0;
#undef MVD
    }
  }
  l->bs[dir][zq] = (uint8_t)bs;
}

/* xPelFilterLuma: TComLoopFilter.cpp:800-859 (no PCM / lossless sides) */
static void pel_filter_luma(int16_t* s, int off, int tc, int sw, int thr_cut, int filt_p, int filt_q, int maxv)
{
  const int m4 = s[0], m3 = s[-off], m5 = s[off], m2 = s[-off * 2], m6 = s[off * 2], m1 = s[-off * 3], m7 = s[off * 3], m0 = s[-off * 4];
  if (sw)
  {
    s[-off] = (int16_t)CLIP3(m3 - 2 * tc, m3 + 2 * tc, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3));
    s[0] = (int16_t)CLIP3(m4 - 2 * tc, m4 + 2 * tc, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3));
    s[-off * 2] = (int16_t)CLIP3(m2 - 2 * tc, m2 + 2 * tc, ((m1 + m2 + m3 + m4 + 2) >> 2));
    s[off] = (int16_t)CLIP3(m5 - 2 * tc, m5 + 2 * tc, ((m3 + m4 + m5 + m6 + 2) >> 2));
    s[-off * 3] = (int16_t)CLIP3(m1 - 2 * tc, m1 + 2 * tc, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3));
    s[off * 2] = (int16_t)CLIP3(m6 - 2 * tc, m6 + 2 * tc, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3));
  }
  else
  {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (iabs(delta) < thr_cut)
    {
      const int tc2 = tc >> 1;
      delta = CLIP3(-tc, tc, delta);
      s[-off] = (int16_t)CLIP3(0, maxv, m3 + delta);
      s[0] = (int16_t)CLIP3(0, maxv, m4 - delta);
      if (filt_p)
      {
        const int d1 = CLIP3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1));
        s[-off * 2] = (int16_t)CLIP3(0, maxv, m2 + d1);
      }
      if (filt_q)
      {
        const int d2 = CLIP3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1));
        s[off] = (int16_t)CLIP3(0, maxv, m5 + d2);
      }
    }
  }
}
static int calc_dp(const int16_t* s, int off) { return iabs(s[-off * 3] - 2 * s[-off * 2] + s[-off]); }
static int calc_dq(const int16_t* s, int off) { return iabs(s[0] - 2 * s[off] + s[off * 2]); }
static int use_strong(int off, int d, int beta, int tc, const int16_t* s)
{
  const int m4 = s[0], m3 = s[-off], m7 = s[off * 3], m0 = s[-off * 4];
  const int d_strong = iabs(m0 - m3) + iabs(m7 - m4);
  return (d_strong < (beta >> 3)) && (d < (beta >> 2)) && (iabs(m3 - m4) < ((tc * 5 + 1) >> 1));
}

/* xEdgeFilterLuma: TComLoopFilter.cpp:540-653 */
static void edge_filter_luma(lf_ctx* l, int ctu, int z_cu, int depth, int dir, int edge)
{
  const geom* g = l->g;
  const int stride = g->w[0];
  const int num_parts = g->pw >> depth;
  const int x0 = zx(z_cu), y0 = zy(z_cu);
  const hmgpu_slice_params* sl = slice_of(g, ctu);
  int16_t* base = l->pic->plane[0] + (size_t)((ctu / g->ctus_w) * g->ctu + y0 * 4) * stride + (ctu % g->ctus_w) * g->ctu + x0 * 4;
  const int off = dir == 0 ? 1 : stride, step = dir == 0 ? stride : 1;
  const int maxv = (1 << g->bd[0]) - 1;
  int idx;
  base += dir == 0 ? edge * 4 : edge * 4 * stride;
  for (idx = 0; idx < num_parts; idx++)
  {
    const int zq = dir == 0 ? xy2z(x0 + edge, y0 + idx) : xy2z(x0 + idx, y0 + edge);
    const int bs = l->bs[dir][zq];
    if (bs)
    {
      int pctu = 0, zp = 0;
      int qp, qp_p, qp_q, tc, beta, index_tc, index_b, side, thr_cut, i;
      int16_t* s = base + step * (idx * 4);
      neighbour(g, ctu, zq, dir, &pctu, &zp);
      qp_q = PM(qp, ctu, zq); qp_p = PM(qp, pctu, zp);
      qp = (qp_p + qp_q + 1) >> 1;
      index_tc = CLIP3(0, 53, qp + 2 * (bs - 1) + (sl->tc_offset_div2 << 1));
      index_b = CLIP3(0, 51, qp + (sl->beta_offset_div2 << 1));
      tc = k_tc_table[index_tc] * (1 << (g->bd[0] - 8));
      beta = k_beta_table[index_b] * (1 << (g->bd[0] - 8));
      side = (beta + (beta >> 1)) >> 3;
      thr_cut = tc * 10;
      {
        const int dp0 = calc_dp(s, off), dq0 = calc_dq(s, off), dp3 = calc_dp(s + step * 3, off), dq3 = calc_dq(s + step * 3, off);
        const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
        if (d < beta)
        {
          const int fp = dp < side, fq = dq < side;
          const int sw = use_strong(off, 2 * d0, beta, tc, s) && use_strong(off, 2 * d3, beta, tc, s + step * 3);
          const int np = no_filter(g, pctu, zp), nq = no_filter(g, ctu, zq);             /* bPartPNoFilter / bPartQNoFilter: :629-634 */
          for (i = 0; i < 4; i++)
          {
            int16_t* t = s + step * i;
            int16_t keep[8];
            int k;
            for (k = 0; k < 8; k++) keep[k] = t[(k - 4) * off];
            pel_filter_luma(t, off, tc, sw, thr_cut, fp, fq, maxv);
            if (np) for (k = 0; k < 4; k++) t[(k - 4) * off] = keep[k];                   /* xPelFilterLuma :847-858 */
            if (nq) for (k = 4; k < 8; k++) t[(k - 4) * off] = keep[k];
          }
        }
      }
    }
  }
}

/* xEdgeFilterChroma: TComLoopFilter.cpp:656-785.  `edge` in 4x4 luma partitions from the CU origin; an edge is filtered when it lies on the
 * 8-sample grid of the CHROMA plane (:684-692): every 16 luma samples across a subsampled direction, every 8 otherwise */
static void edge_filter_chroma(lf_ctx* l, int ctu, int z_cu, int depth, int dir, int edge)
{
  const geom* g = l->g;
  const int stride = g->w[1];
  const int num_parts = g->pw >> depth;
  const int x0 = zx(z_cu), y0 = zy(z_cu);
  const hmgpu_slice_params* sl = slice_of(g, ctu);
  const int off = dir == 0 ? 1 : stride, step = dir == 0 ? stride : 1;
  const int maxv = (1 << g->bd[1]) - 1;
  const int pels_h = 4 >> g->csx[1], pels_v = 4 >> g->csy[1];           /* uiPelsInPartChromaH / V */
  const int loop = dir == 0 ? pels_v : pels_h;                           /* samples along the edge per partition */
  int idx, comp;
  if ((dir == 0 && ((x0 + edge) % (8 / pels_h))) || (dir == 1 && ((y0 + edge) % (8 / pels_v)))) return;
  for (idx = 0; idx < num_parts; idx++)
  {
    const int zq = dir == 0 ? xy2z(x0 + edge, y0 + idx) : xy2z(x0 + idx, y0 + edge);
    const int bs = l->bs[dir][zq];
    if (bs > 1)
    {
      int pctu = 0, zp = 0, qp_p, qp_q;
      neighbour(g, ctu, zq, dir, &pctu, &zp);
      qp_q = PM(qp, ctu, zq); qp_p = PM(qp, pctu, zp);
      for (comp = 1; comp < 3; comp++)
      {
        int16_t* base = l->pic->plane[comp] + (size_t)((ctu / g->ctus_w) * (g->ctu >> g->csy[1]) + y0 * pels_v) * stride + (ctu % g->ctus_w) * (g->ctu >> g->csx[1]) + x0 * pels_h;
        int qp = ((qp_p + qp_q + 1) >> 1) + (comp == 1 ? sl->pps_cb_qp_offset : sl->pps_cr_qp_offset);
        int index_tc, tc, stp;
        if (qp >= 58) { if (g->fmt == 1) qp -= 6; else if (qp > 51) qp = 51; }                        /* :761-765 */
        else if (qp >= 0) qp = scaled_chroma_qp(qp, g->fmt);
        index_tc = CLIP3(0, 53, qp + 2 * (bs - 1) + (sl->tc_offset_div2 << 1));
        tc = k_tc_table[index_tc] * (1 << (g->bd[1] - 8));
        base += dir == 0 ? edge * pels_h : edge * pels_v * stride;
        for (stp = 0; stp < loop; stp++)
        {
          /* xPelFilterChroma: :870-891 */
          int16_t* s = base + step * (stp + idx * loop);
          const int m4 = s[0], m3 = s[-off], m5 = s[off], m2 = s[-off * 2];
          const int delta = CLIP3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
          if (!no_filter(g, pctu, zp)) s[-off] = (int16_t)CLIP3(0, maxv, m3 + delta);      /* xPelFilterChroma :883-890 */
          if (!no_filter(g, ctu, zq)) s[0] = (int16_t)CLIP3(0, maxv, m4 - delta);
        }
      }
    }
  }
}

/* xDeblockCU: TComLoopFilter.cpp:167-234.  filter != 0: apply the sample filters; bs_out != NULL: export final Bs */
static void deblock_cu(lf_ctx* l, int ctu, int z, int depth, int dir, int filter, uint8_t* bs_out)
{
  const geom* g = l->g;
  const int cur_parts = g->parts >> (2 * depth);
  if (PM(part_size, ctu, z) == HMGPU_SIZE_NONE) return;
  if (PM(depth, ctu, z) > depth)
  {
    const int q = cur_parts >> 2;
    int i;
    for (i = 0; i < 4; i++)
    {
      const int zi = z + i * q;
      const int x = (ctu % g->ctus_w) * g->ctu + zx(zi) * 4, y = (ctu / g->ctus_w) * g->ctu + zy(zi) * 4;
      if (x < g->seq->width && y < g->seq->height) deblock_cu(l, ctu, zi, depth + 1, dir, filter, bs_out);
    }
    return;
  }
  {
    /* xSetLoopfilterParam: :356-409 */
    const hmgpu_slice_params* sl = slice_of(g, ctu);
    const int x = (ctu % g->ctus_w) * g->ctu + zx(z) * 4, y = (ctu / g->ctus_w) * g->ctu + zy(z) * 4;
    const int size_pu = g->pw >> depth;
    int nc, nz, p, e;
    l->internal_edge = !sl->deblocking_disable;
    l->left_edge = (x != 0) && !sl->deblocking_disable && neighbour(g, ctu, z, 0, &nc, &nz);
    l->top_edge = (y != 0) && !sl->deblocking_disable && neighbour(g, ctu, z, 1, &nc, &nz);
    set_edge_tu(l, ctu, z, 0, g->seq->log2_ctu_size - depth);
    /* xSetEdgefilterPU: :293-353 */
    set_edge_cu(l, z, size_pu, 0, 0, l->left_edge);
    set_edge_cu(l, z, size_pu, 1, 0, l->top_edge);
    switch (PM(part_size, ctu, z))
    {
      case HMGPU_SIZE_2NxN:  set_edge_cu(l, z, size_pu, 1, size_pu >> 1, l->internal_edge); break;
      case HMGPU_SIZE_Nx2N:  set_edge_cu(l, z, size_pu, 0, size_pu >> 1, l->internal_edge); break;
      case HMGPU_SIZE_NxN:   set_edge_cu(l, z, size_pu, 0, size_pu >> 1, l->internal_edge);
                             set_edge_cu(l, z, size_pu, 1, size_pu >> 1, l->internal_edge); break;
      case HMGPU_SIZE_2NxnU: set_edge_cu(l, z, size_pu, 1, size_pu >> 2, l->internal_edge); break;
      case HMGPU_SIZE_2NxnD: set_edge_cu(l, z, size_pu, 1, size_pu - (size_pu >> 2), l->internal_edge); break;
      case HMGPU_SIZE_nLx2N: set_edge_cu(l, z, size_pu, 0, size_pu >> 2, l->internal_edge); break;
      case HMGPU_SIZE_nRx2N: set_edge_cu(l, z, size_pu, 0, size_pu - (size_pu >> 2), l->internal_edge); break;
      default: break;
    }
    for (p = z; p < z + cur_parts; p++)
    {
      /* Bs only on the 8x8 grid (uiBSCheck, :199-206) */
      const int check = (dir == 0 && (p % 2) == 0) || (dir == 1 && ((p % 4) / 2) == 0);
      if (l->edge[dir][p] && check) boundary_strength(l, ctu, dir, p);
    }
    for (e = 0; e < size_pu; e += 2)
    {
      if (bs_out)
      {
        int u;
        for (u = 0; u < size_pu; u++)
        {
          const int zq = dir == 0 ? xy2z(zx(z) + e, zy(z) + u) : xy2z(zx(z) + u, zy(z) + e);
          bs_out[(size_t)ctu * g->parts + zq] = l->bs[dir][zq];
        }
      }
      if (filter)
      {
        edge_filter_luma(l, ctu, z, depth, dir, e);
        /* (HM calls it where (uiPelsInPart >= DEBLOCK_SMALLEST_BLOCK) || the edge index is a multiple of two chroma partitions,
         * TComLoopFilter.cpp:225-229; the function's own test on the edge's position in the CTU decides) */
        if (g->seq->chroma_format != 0) edge_filter_chroma(l, ctu, z, depth, dir, e);
      }
    }
  }
}

/* loopFilterPic: TComLoopFilter.cpp:130-155 */
static int loop_filter(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                       const hmgpu_pic_params* pp, hmo_picture* pic, int dir_mask, uint8_t* bs_ver, uint8_t* bs_hor)
{
  geom g;
  lf_ctx* l = (lf_ctx*)malloc(sizeof(lf_ctx));
  int dir, a;
  if (!l) return HMGPU_ENOMEM;
  geom_init(&g, seq, slices, meta, pp);
  l->g = &g; l->pic = pic;
  for (dir = 0; dir < 2; dir++)
  {
    if (!((dir_mask >> dir) & 1)) continue;
    for (a = 0; a < g.num_ctus; a++)
    {
      memset(l->bs[dir], 0, sizeof(l->bs[dir]));
      memset(l->edge[dir], 0, sizeof(l->edge[dir]));
      deblock_cu(l, a, 0, 0, dir, pic != NULL, dir == 0 ? bs_ver : bs_hor);
    }
  }
  free(l);
  return HMGPU_OK;
}

int hmo_loop_filter_pic(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                        const hmgpu_pic_params* pp, hmo_picture* pic, int dir_mask)
{
  return loop_filter(seq, slices, meta, pp, pic, dir_mask, NULL, NULL);
}

int hmo_boundary_strengths(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_ctu_meta* meta,
                           const hmgpu_pic_params* pp, uint8_t* bs_ver, uint8_t* bs_hor)
{
  const size_t n = (size_t)hmgpu_num_ctus(seq) * hmgpu_parts_per_ctu(seq);
  memset(bs_ver, 0, n); memset(bs_hor, 0, n);
  return loop_filter(seq, slices, meta, pp, NULL, 3, bs_ver, bs_hor);
}

/* geometry helpers of the ABI, restated for the oracle library (the product has its own) */
int32_t hmgpu_num_ctus(const hmgpu_seq_params* seq)
{
  const int c = 1 << seq->log2_ctu_size;
  return ((seq->width + c - 1) / c) * ((seq->height + c - 1) / c);
}
int32_t hmgpu_parts_per_ctu(const hmgpu_seq_params* seq) { return 1 << (2 * seq->log2_ctu_size - 4); }

/* ------------------------------------------------------------------------------------------------ SAO */
/* reconstructBlkSAOParams + reconstructBlkSAOParam + invertQuantOffsets + getMergeList:
 * TComSampleAdaptiveOffset.cpp:229-372; merge availability TComPic.cpp:132-137 */
int hmo_sao_reconstruct_params(const hmgpu_seq_params* seq, const hmgpu_pic_params* pp, const hmgpu_ctu_meta* meta,
                               const hmgpu_sao_param* raw, hmgpu_sao_param* rec)
{
  geom g;
  int a, comp, i;
  geom_init(&g, seq, NULL, meta, pp);
  memcpy(rec, raw, sizeof(hmgpu_sao_param) * 3 * g.num_ctus);
  for (a = 0; a < g.num_ctus; a++)
  {
    const int cx = a % g.ctus_w, cy = a / g.ctus_w;
    const hmgpu_sao_param* merge[2] = { NULL, NULL };       /* [SAO_MERGE_LEFT], [SAO_MERGE_ABOVE] */
    if (cy > 0 && slice_id(&g, a - g.ctus_w) == slice_id(&g, a) && tile_id(&g, a - g.ctus_w) == tile_id(&g, a))
      merge[HMGPU_SAO_MERGE_ABOVE] = &rec[(size_t)(a - g.ctus_w) * 3];
    if (cx > 0 && slice_id(&g, a - 1) == slice_id(&g, a) && tile_id(&g, a - 1) == tile_id(&g, a))
      merge[HMGPU_SAO_MERGE_LEFT] = &rec[(size_t)(a - 1) * 3];
    for (comp = 0; comp < 3; comp++)
    {
      hmgpu_sao_param* p = &rec[(size_t)a * 3 + comp];
      const int shift = comp == 0 ? pp->sao_offset_shift_luma : pp->sao_offset_shift_chroma;
      if (p->mode_idc == HMGPU_SAO_OFF) continue;
      if (p->mode_idc == HMGPU_SAO_NEW)
      {
        int32_t coded[32];
        memcpy(coded, p->offset, sizeof(coded));
        memset(p->offset, 0, sizeof(p->offset));
        if (p->type_idc == HMGPU_SAO_BO)
          for (i = 0; i < 4; i++) p->offset[(p->type_aux_info + i) % 32] = coded[(p->type_aux_info + i) % 32] * (1 << shift);
        else
          for (i = 0; i < 5; i++) p->offset[i] = coded[i] * (1 << shift);
      }
      else
      {
        const hmgpu_sao_param* t = merge[p->type_idc];
        if (!t) return HMGPU_EINVAL;                          /* HM: assert(mergeTarget != NULL) */
        *p = t[comp];
      }
    }
  }
  return HMGPU_OK;
}

static int sgn(int v) { return (v > 0) - (v < 0); }

/* offsetBlock: TComSampleAdaptiveOffset.cpp:375-661, restated in HM's own loop shape (line buffers replaced by direct
 * recomputation of the sign terms, which is value-identical) */
void hmo_sao_offset_block(int bit_depth, int type_idx, const int32_t* offset, const int16_t* src, int16_t* res,
                          int src_stride, int res_stride, int w, int h, const int32_t* avail)
{
  const int maxv = (1 << bit_depth) - 1;
  const int L = avail[0], R = avail[1], A = avail[2], B = avail[3], AL = avail[4], AR = avail[5], BL = avail[6], BR = avail[7];
  int x, y;
#define S(xx, yy) ((int)src[(yy) * src_stride + (xx)])
#define OUT(xx, yy, et) res[(yy) * res_stride + (xx)] = (int16_t)CLIP3(0, maxv, S(xx, yy) + offset[2 + (et)])
  switch (type_idx)
  {
    case HMGPU_SAO_EO_0:
    {
      const int sx = L ? 0 : 1, ex = R ? w : w - 1;
      for (y = 0; y < h; y++) for (x = sx; x < ex; x++) OUT(x, y, sgn(S(x, y) - S(x - 1, y)) + sgn(S(x, y) - S(x + 1, y)));
      break;
    }
    case HMGPU_SAO_EO_90:
    {
      const int sy = A ? 0 : 1, ey = B ? h : h - 1;
      for (y = sy; y < ey; y++) for (x = 0; x < w; x++) OUT(x, y, sgn(S(x, y) - S(x, y - 1)) + sgn(S(x, y) - S(x, y + 1)));
      break;
    }
    case HMGPU_SAO_EO_135:
    {
      const int sx = L ? 0 : 1, ex = R ? w : w - 1;
      const int f0 = AL ? 0 : 1, f1 = A ? ex : 1;
      const int l0 = B ? sx : w - 1, l1 = BR ? w : w - 1;
      for (x = f0; x < f1; x++) OUT(x, 0, sgn(S(x, 0) - S(x - 1, -1)) + sgn(S(x, 0) - S(x + 1, 1)));
      for (y = 1; y < h - 1; y++) for (x = sx; x < ex; x++) OUT(x, y, sgn(S(x, y) - S(x - 1, y - 1)) + sgn(S(x, y) - S(x + 1, y + 1)));
      for (x = l0; x < l1; x++) OUT(x, h - 1, sgn(S(x, h - 1) - S(x - 1, h - 2)) + sgn(S(x, h - 1) - S(x + 1, h)));
      break;
    }
    case HMGPU_SAO_EO_45:
    {
      const int sx = L ? 0 : 1, ex = R ? w : w - 1;
      const int f0 = A ? sx : w - 1, f1 = AR ? w : w - 1;
      const int l0 = BL ? 0 : 1, l1 = B ? ex : 1;
      for (x = f0; x < f1; x++) OUT(x, 0, sgn(S(x, 0) - S(x + 1, -1)) + sgn(S(x, 0) - S(x - 1, 1)));
      for (y = 1; y < h - 1; y++) for (x = sx; x < ex; x++) OUT(x, y, sgn(S(x, y) - S(x + 1, y - 1)) + sgn(S(x, y) - S(x - 1, y + 1)));
      for (x = l0; x < l1; x++) OUT(x, h - 1, sgn(S(x, h - 1) - S(x + 1, h - 2)) + sgn(S(x, h - 1) - S(x - 1, h)));
      break;
    }
    case HMGPU_SAO_BO:
    {
      const int shift = bit_depth - 5;
      for (y = 0; y < h; y++) for (x = 0; x < w; x++) res[y * res_stride + x] = (int16_t)CLIP3(0, maxv, S(x, y) + offset[S(x, y) >> shift]);
      break;
    }
    default: break;
  }
#undef S
#undef OUT
}

/* TComPicSym::deriveLoopFilterBoundaryAvailibility: TComPicSym.cpp:365-471 */
static void sao_avail(const geom* g, int a, int32_t av[8])
{
  const int cx = a % g->ctus_w, cy = a / g->ctus_w;
  const int dx[8] = { -1, 1, 0, 0, -1, 1, -1, 1 }, dy[8] = { 0, 0, -1, 1, -1, -1, 1, 1 };
  int k;
  for (k = 0; k < 8; k++)
  {
    const int nx = cx + dx[k], ny = cy + dy[k];
    av[k] = nx >= 0 && nx < g->ctus_w && ny >= 0 && ny < g->ctus_h;
    if (av[k])
    {
      const int n = ny * g->ctus_w + nx;
      const int sc = slice_id(g, a), sn = slice_id(g, n);
      if (sc != sn)
      {
        /* the flag of whichever of the two slices comes later in decoding order decides (left/above/aboveLeft: current;
         * right/below/belowRight: the neighbour; aboveRight/belowLeft: the later one) */
        const hmgpu_slice_params* later = sc > sn ? &g->slices[sc] : &g->slices[sn];
        av[k] = later->lf_across_slices != 0;
      }
      if (g->pp && !g->pp->lf_across_tiles && av[k]) av[k] = tile_id(g, n) == tile_id(g, a);
    }
  }
}

/* SAOProcess + offsetCTU: TComSampleAdaptiveOffset.cpp:663-734 */
int hmo_sao_process(const hmgpu_seq_params* seq, const hmgpu_slice_params* slices, const hmgpu_pic_params* pp,
                    const hmgpu_ctu_meta* meta, const hmgpu_sao_param* rec, const hmo_picture* src, hmo_picture* dst)
{
  geom g;
  int a, comp;
  geom_init(&g, seq, slices, meta, pp);
  for (a = 0; a < g.num_ctus; a++)
  {
    int32_t av[8];
    const int yp = (a / g.ctus_w) * g.ctu, xp = (a % g.ctus_w) * g.ctu;
    const int hh = imin(g.ctu, seq->height - yp), ww = imin(g.ctu, seq->width - xp);
    int all_off = 1;
    for (comp = 0; comp < 3; comp++) if (rec[(size_t)a * 3 + comp].mode_idc != HMGPU_SAO_OFF) all_off = 0;
    if (all_off) continue;
    sao_avail(&g, a, av);
    for (comp = 0; comp < 3; comp++)
    {
      const hmgpu_sao_param* p = &rec[(size_t)a * 3 + comp];
      const int sx = g.csx[comp], sy = g.csy[comp];
      if (p->mode_idc == HMGPU_SAO_OFF) continue;
      hmo_sao_offset_block(g.bd[comp], p->type_idc, p->offset,
                           src->plane[comp] + (size_t)(yp >> sy) * g.w[comp] + (xp >> sx),
                           dst->plane[comp] + (size_t)(yp >> sy) * g.w[comp] + (xp >> sx),
                           g.w[comp], g.w[comp], ww >> sx, hh >> sy, av);
    }
  }
  /* PCMLFDisableProcess (TComSampleAdaptiveOffset.cpp:742-835): PCM (filter disabled) and lossless CUs get their
   * reconstruction back; deblocking never touched them, so the SAO input already holds it */
  for (a = 0; a < g.num_ctus; a++)
  {
    int z;
    for (z = 0; z < g.parts; z++)
    {
      const int px = (a % g.ctus_w) * g.ctu + zx(z) * 4, py = (a / g.ctus_w) * g.ctu + zy(z) * 4;
      if (px >= seq->width || py >= seq->height || !no_filter(&g, a, z)) continue;
      for (comp = 0; comp < 3; comp++)
      {
        const int sx = g.csx[comp], sy = g.csy[comp];
        int x, y;
        for (y = 0; y < (4 >> sy); y++)
          for (x = 0; x < (4 >> sx); x++)
          {
            const size_t o = (size_t)((py >> sy) + y) * g.w[comp] + (px >> sx) + x;
            dst->plane[comp][o] = src->plane[comp][o];
          }
      }
    }
  }
  return HMGPU_OK;
}
