/* oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" harness around the REAL HM 16.0 libraries (compiled by oracle/Makefile from
 * the sources where they lie under /root/reference; nothing of HM is copied here).  It is used
 *   (1) by oracle/make_golden.py to manufacture the fixtures under tests/golden/ and
 *   (2) by tests/ (when oracle/_ref/libhmref.so is present) to check the C restatement
 *       (oracle/hm_oracle.c) and, optionally, as the "reference" CPU baseline in bench.py.
 * The product (libhm_amd/) never links, loads or calls it.
 *
 * Two groups of entry points:
 *   ref_kat_*   direct calls of HM's own free/public functions on caller-supplied arrays
 *               (xITrMxN, TComInterpolationFilter::filterHor/Ver, TComYuv::addAvg,
 *                TComSampleAdaptiveOffset::offsetBlock)
 *   ref_dec_*   drive HM's TDecTop over an Annex-B stream exactly like TAppDecTop::decode
 *               (TAppDecTop.cpp:93-215), but stop between decompressSlice() and filterPicture()
 *               so that the per-CTU TComDataCU metadata, the coefficients and the planes
 *               before deblocking / after deblocking / after SAO can be read out.  The filter
 *               stage re-states the call sequence of TDecGop::filterPicture (TDecGop.cpp:157-217)
 *               and TDecTop::executeLoopFilters (TDecTop.cpp:192-213) using HM's own objects.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <list>
#include <map>
#include <string>
#include <sstream>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <limits>

/* the harness needs HM's internals (m_pcPic, m_cLoopFilter, offsetBlock ...) */
#define private public
#define protected public
#include "TLibCommon/CommonDef.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComInterpolationFilter.h"
#include "TLibCommon/TComYuv.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComLoopFilter.h"
#include "TLibCommon/TComSampleAdaptiveOffset.h"
#include "TLibDecoder/TDecTop.h"
#include "TLibDecoder/NALread.h"
#undef private
#undef protected

extern Void xITrMxN(Int bitDepth, TCoeff *coeff, TCoeff *block, Int iWidth, Int iHeight, Bool useDST, const Int maxTrDynamicRange);
Bool g_md5_mismatch = false;   /* HM expects the application to define it (decmain.cpp) */

static bool g_romReady = false;

static void set_globals(int bdY, int bdC)
{
  if (!g_romReady) { initROM(); g_romReady = true; }
  g_bitDepth[CHANNEL_TYPE_LUMA] = bdY;
  g_bitDepth[CHANNEL_TYPE_CHROMA] = bdC;
  g_maxTrDynamicRange[CHANNEL_TYPE_LUMA] = 15;
  g_maxTrDynamicRange[CHANNEL_TYPE_CHROMA] = 15;
}

extern "C" {

/* ------------------------------------------------------------------ KATs --------------------------- */

int ref_kat_init(int bdY, int bdC) { set_globals(bdY, bdC); return 0; }

/* HM xITrMxN (TComTrQuant.cpp:894): n TUs of w x h, int32 in / int32 out */
void ref_kat_itr(int bitDepth, const int32_t* coeff, int32_t* block, int w, int h, int useDST, int n)
{
  std::vector<TCoeff> c(w * h);
  for (int i = 0; i < n; i++)
  {
    memcpy(&c[0], coeff + (size_t)i * w * h, sizeof(TCoeff) * w * h);   /* xITrMxN may not modify, but be safe */
    xITrMxN(bitDepth, &c[0], block + (size_t)i * w * h, w, h, useDST != 0, 15);
  }
}

/* one prediction block the way TComPrediction::xPredInterBlk (TComPrediction.cpp:660-698) dispatches
 * to TComInterpolationFilter.  src points at the integer-sample position of the block inside a plane
 * that has at least 4 samples of margin on every side.  comp: 0 luma, 1 chroma (4:2:0).            */
void ref_kat_interp(int comp, const int16_t* src, int srcStride, int16_t* dst, int dstStride,
                    int w, int h, int xFrac, int yFrac, int bi)
{
  TComInterpolationFilter f;
  const ComponentID compID = comp ? COMPONENT_Cb : COMPONENT_Y;
  const ChromaFormat fmt = CHROMA_420;
  Pel* ref = const_cast<Pel*>(src);
  if (yFrac == 0)
    f.filterHor(compID, ref, srcStride, dst, dstStride, w, h, xFrac, !bi, fmt);
  else if (xFrac == 0)
    f.filterVer(compID, ref, srcStride, dst, dstStride, w, h, yFrac, true, !bi, fmt);
  else
  {
    const int vFilterSize = comp ? NTAPS_CHROMA : NTAPS_LUMA;
    const int tmpStride = w;
    std::vector<Pel> tmp((size_t)tmpStride * (h + vFilterSize - 1));
    f.filterHor(compID, ref - ((vFilterSize >> 1) - 1) * srcStride, srcStride, &tmp[0], tmpStride, w, h + vFilterSize - 1, xFrac, false, fmt);
    f.filterVer(compID, &tmp[0] + ((vFilterSize >> 1) - 1) * tmpStride, tmpStride, dst, dstStride, w, h, yFrac, false, !bi, fmt);
  }
}

/* HM TComYuv::addAvg (TComYuv.cpp:336) on the luma plane of w x h scratch blocks */
void ref_kat_addavg(const int16_t* s0, const int16_t* s1, int16_t* dst, int w, int h)
{
  TComYuv a, b, d;
  a.create(w, h, CHROMA_400); b.create(w, h, CHROMA_400); d.create(w, h, CHROMA_400);
  for (int y = 0; y < h; y++)
  {
    memcpy(a.getAddr(COMPONENT_Y) + y * a.getStride(COMPONENT_Y), s0 + y * w, w * sizeof(Pel));
    memcpy(b.getAddr(COMPONENT_Y) + y * b.getStride(COMPONENT_Y), s1 + y * w, w * sizeof(Pel));
  }
  d.addAvg(&a, &b, 0, w, h);
  for (int y = 0; y < h; y++)
    memcpy(dst + y * w, d.getAddr(COMPONENT_Y) + y * d.getStride(COMPONENT_Y), w * sizeof(Pel));
  a.destroy(); b.destroy(); d.destroy();
}

/* HM TComSampleAdaptiveOffset::offsetBlock (TComSampleAdaptiveOffset.cpp:375) on one block.
 * src/res point at the block's top-left sample inside planes with >= 1 sample of margin.
 * avail[8] = left,right,above,below,aboveLeft,aboveRight,belowLeft,belowRight.  offset[32] as after
 * reconstructBlkSAOParams.  comp: 0..2                                                              */
void ref_kat_sao_block(int comp, int bdY, int bdC, int typeIdx, const int32_t* offset,
                       const int16_t* src, int16_t* res, int srcStride, int resStride, int w, int h, const int32_t* avail)
{
  set_globals(bdY, bdC);
  TComSampleAdaptiveOffset sao;
  sao.create(64, 64, CHROMA_420, 64, 64, 4, 0, 0);
  Int off[MAX_NUM_SAO_CLASSES];
  for (int i = 0; i < MAX_NUM_SAO_CLASSES; i++) off[i] = offset[i];
  sao.offsetBlock(ComponentID(comp), typeIdx, off, const_cast<Pel*>(src), res, srcStride, resStride, w, h,
                  avail[0] != 0, avail[1] != 0, avail[2] != 0, avail[3] != 0, avail[4] != 0, avail[5] != 0, avail[6] != 0, avail[7] != 0);
  sao.destroy();
}

/* ------------------------------------------------------------------ stream decode ------------------ */

struct RefDec
{
  TDecTop            top;
  std::vector<uint8_t> bs;
  std::vector<std::pair<size_t, size_t> > nals;   /* payload [begin,end) of each NAL unit */
  size_t             nextNal;
  Int                pocLastDisplay;
  Int                skipFrame;
  bool               pending;       /* a picture is reconstructed and waits for the filter stage */
  bool               atEnd;
  int                stage;         /* 0 = before deblock, 1 = after deblock, 2 = after SAO */
  int                numMismatch;
  RefDec() : nextNal(0), pocLastDisplay(-MAX_INT), skipFrame(0), pending(false), atEnd(false), stage(0), numMismatch(0) {}
};

static void split_nals(RefDec* d)
{
  const std::vector<uint8_t>& b = d->bs;
  std::vector<size_t> starts;   /* index of first payload byte after a 00 00 01 */
  for (size_t i = 0; i + 2 < b.size(); i++)
    if (b[i] == 0 && b[i + 1] == 0 && b[i + 2] == 1) { starts.push_back(i + 3); i += 2; }
  for (size_t k = 0; k < starts.size(); k++)
  {
    size_t end = (k + 1 < starts.size()) ? starts[k + 1] - 3 : b.size();
    while (end > starts[k] && b[end - 1] == 0) end--;          /* trailing_zero_8bits / leading zero of next start code */
    d->nals.push_back(std::make_pair(starts[k], end));
  }
}

void* ref_dec_open(const uint8_t* data, int64_t len, int checkHash)
{
  RefDec* d = new RefDec();
  d->bs.assign(data, data + len);
  split_nals(d);
  d->top.create();
  d->top.init();
  d->top.setDecodedPictureHashSEIEnabled(checkHash);
  g_md5_mismatch = false;
  return d;
}

void ref_dec_close(void* h)
{
  RefDec* d = (RefDec*)h;
  d->top.deletePicBuffer();
  d->top.destroy();
  delete d;
}

/* decode NAL units until a whole picture has been reconstructed (all its slices went through
 * TDecGop::decompressSlice) but not yet filtered.  returns 1 if such a picture is pending, 0 at end */
int ref_dec_next(void* h)
{
  RefDec* d = (RefDec*)h;
  if (d->pending) return 1;
  while (d->nextNal < d->nals.size())
  {
    const std::pair<size_t, size_t>& r = d->nals[d->nextNal];
    std::vector<uint8_t> nalUnit(d->bs.begin() + r.first, d->bs.begin() + r.second);
    InputNALUnit nalu;
    read(nalu, nalUnit);
    if (getenv("HMREF_DEBUG")) fprintf(stderr, "[hmref] nal %zu type %d size %zu\n", d->nextNal, (int)nalu.m_nalUnitType, nalUnit.size());
    Bool bNewPicture = d->top.decode(nalu, d->skipFrame, d->pocLastDisplay);
    if (getenv("HMREF_DEBUG")) fprintf(stderr, "[hmref]   -> newPicture %d\n", (int)bNewPicture);
    if (!bNewPicture) d->nextNal++;                  /* else: same NAL is pushed again (TAppDecTop.cpp:168-182) */
    if ((bNewPicture || nalu.m_nalUnitType == NAL_UNIT_EOS) && d->top.m_pcPic && !d->top.m_bFirstSliceInPicture)
    {
      d->pending = true; d->stage = 0;
      return 1;
    }
  }
  if (!d->atEnd)
  {
    d->atEnd = true;
    if (d->top.m_pcPic && !d->top.m_bFirstSliceInPicture) { d->pending = true; d->stage = 0; return 1; }
  }
  return 0;
}

static TComPic* cur(RefDec* d) { return d->top.m_pcPic; }

/* info[]: 0 width 1 height 2 bdY 3 bdC 4 POC 5 sliceType(of slice 0) 6 numCTUs 7 ctusInWidth 8 partsPerCTU
 *         9 maxCUWidth 10 numSlices 11 useSAO 12 lfAcrossTiles 13 chromaFormat 14 temporalId 15 maxCUDepth(total) */
void ref_dec_info(void* h, int32_t* info)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d); TComSlice* s = p->getSlice(0);
  info[0] = s->getSPS()->getPicWidthInLumaSamples();
  info[1] = s->getSPS()->getPicHeightInLumaSamples();
  info[2] = g_bitDepth[CHANNEL_TYPE_LUMA];
  info[3] = g_bitDepth[CHANNEL_TYPE_CHROMA];
  info[4] = s->getPOC();
  info[5] = (int)s->getSliceType();
  info[6] = p->getNumCUsInFrame();
  info[7] = p->getFrameWidthInCU();
  info[8] = p->getNumPartInCU();
  info[9] = g_uiMaxCUWidth;
  info[10] = d->top.m_uiSliceIdx;
  info[11] = s->getSPS()->getUseSAO() ? 1 : 0;
  info[12] = s->getPPS()->getLoopFilterAcrossTilesEnabledFlag() ? 1 : 0;
  info[13] = (int)p->getChromaFormat();
  info[14] = s->getTLayer();
  info[15] = g_uiMaxCUDepth;
}

/* per-slice constants, 64 ints per slice:
 * 0 sliceType 1 sliceQp 2 ppsCbOff 3 ppsCrOff 4 sliceCbDelta 5 sliceCrDelta 6 deblockDisable 7 betaOffDiv2 8 tcOffDiv2
 * 9 lfAcrossSlices 10 saoLuma 11 saoChroma 12 numRefIdx0 13 numRefIdx1 14 useWP 15 wpBiPred 16 transquantBypassEnable
 * 17 usePCM 18 pcmFilterDisable 19 sliceCurStartCUAddr(in partitions) 20 scalingListEnabled 21 signHiding(unused)
 * 22 useTransformSkip 23 sliceCurEndCUAddr 24 chromaQpAdjTableSize 25 strongIntraSmoothing 26 constrainedIntraPred 27 flagsValid(=1)
 * 28 range-extension tools as HMGPU_REXT_* bits (rotation 1, implicit RDPCM 2, explicit RDPCM 4, intra smoothing disabled 8) 29 crossComponentPrediction
 * 32..47 refPOC L0, 48..63 refPOC L1                                                                               */
void ref_dec_slices(void* h, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  for (UInt i = 0; i < d->top.m_uiSliceIdx; i++)
  {
    TComSlice* s = p->getSlice(i); int32_t* o = out + 64 * i;
    memset(o, 0, 64 * sizeof(int32_t));
    o[0] = (int)s->getSliceType(); o[1] = s->getSliceQp();
    o[2] = s->getPPS()->getQpOffset(COMPONENT_Cb); o[3] = s->getPPS()->getQpOffset(COMPONENT_Cr);
    o[4] = s->getSliceChromaQpDelta(COMPONENT_Cb); o[5] = s->getSliceChromaQpDelta(COMPONENT_Cr);
    o[6] = s->getDeblockingFilterDisable() ? 1 : 0;
    o[7] = s->getDeblockingFilterBetaOffsetDiv2(); o[8] = s->getDeblockingFilterTcOffsetDiv2();
    o[9] = s->getLFCrossSliceBoundaryFlag() ? 1 : 0;
    o[10] = s->getSaoEnabledFlag(CHANNEL_TYPE_LUMA) ? 1 : 0; o[11] = s->getSaoEnabledFlag(CHANNEL_TYPE_CHROMA) ? 1 : 0;
    o[12] = s->isIntra() ? 0 : s->getNumRefIdx(REF_PIC_LIST_0);
    o[13] = s->isInterB() ? s->getNumRefIdx(REF_PIC_LIST_1) : 0;
    o[14] = s->getPPS()->getUseWP() ? 1 : 0; o[15] = s->getPPS()->getWPBiPred() ? 1 : 0;
    o[16] = s->getPPS()->getTransquantBypassEnableFlag() ? 1 : 0;
    o[17] = s->getSPS()->getUsePCM() ? 1 : 0; o[18] = s->getSPS()->getPCMFilterDisableFlag() ? 1 : 0;
    o[19] = s->getSliceCurStartCUAddr();
    o[20] = s->getSPS()->getScalingListFlag() ? 1 : 0;
    o[22] = s->getPPS()->getUseTransformSkip() ? 1 : 0;
    o[23] = s->getSliceCurEndCUAddr();
    o[25] = s->getSPS()->getUseStrongIntraSmoothing() ? 1 : 0;
    o[26] = s->getPPS()->getConstrainedIntraPred() ? 1 : 0;
    o[27] = 1;
    o[28] = (s->getSPS()->getUseResidualRotation() ? 1 : 0) | (s->getSPS()->getUseResidualDPCM(RDPCM_SIGNAL_IMPLICIT) ? 2 : 0) |
            (s->getSPS()->getUseResidualDPCM(RDPCM_SIGNAL_EXPLICIT) ? 4 : 0) | (s->getSPS()->getDisableIntraReferenceSmoothing() ? 8 : 0);
    o[29] = s->getPPS()->getUseCrossComponentPrediction() ? 1 : 0;
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < o[12 + l] && r < 16; r++)
        o[32 + 16 * l + r] = s->getRefPOC(RefPicList(l), r);
  }
}

/* explicit weighted prediction tables after initWpScaling (TComSlice.cpp:1495-1522), 195 ints per slice:
 * 0 applyWP (TComSlice.h:1477), 1 log2WeightDenom luma, 2 chroma, then weight[list][ref 0..15][comp] (96 ints) and
 * offset[list][ref][comp] (96 ints, already scaled to the bit depth as TComWeightPrediction::getWpScaling does, :230-271) */
void ref_dec_wp(void* h, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  for (UInt i = 0; i < d->top.m_uiSliceIdx; i++)
  {
    TComSlice* s = p->getSlice(i); int32_t* o = out + 195 * i;
    memset(o, 0, 195 * sizeof(int32_t));
    o[0] = s->applyWP() ? 1 : 0;
    if (!o[0]) continue;
    const Bool hp = s->getSPS()->getUseHighPrecisionPredictionWeighting();
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < 16 && r < MAX_NUM_REF; r++)
      {
        WPScalingParam* wp; s->getWpScaling(RefPicList(l), r, wp);
        for (int c = 0; c < 3; c++)
        {
          if (l == 0 && r == 0) o[1 + (c ? 1 : 0)] = wp[c].uiLog2WeightDenom;
          o[3 + (l * 16 + r) * 3 + c] = wp[c].iWeight;
          o[99 + (l * 16 + r) * 3 + c] = wp[c].iOffset * (hp ? 1 : (1 << (g_bitDepth[c ? CHANNEL_TYPE_CHROMA : CHANNEL_TYPE_LUMA] - 8)));
        }
      }
  }
}

/* TComPicSym::getTileIdxMap per CTU (raster address) */
void ref_dec_tile_idx(void* h, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  for (UInt a = 0; a < p->getNumCUsInFrame(); a++) out[a] = (int32_t)p->getPicSym()->getTileIdxMap(a);
}

/* scaling lists of the picture's first slice as TDecTop activated them (TDecTop.cpp:651-668): out[0] = enabled, then
 * coef[sizeId 0..3][listId 0..5][64] (TComScalingList::getScalingListAddress, raster order, first 16 used for 4x4) and
 * dc[sizeId][listId] (getScalingListDC) */
void ref_dec_scaling_lists(void* h, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d); TComSlice* s = p->getSlice(0);
  memset(out, 0, (1 + 4 * 6 * 64 + 4 * 6) * sizeof(int32_t));
  if (!s->getSPS()->getScalingListFlag()) return;
  out[0] = 1;
  TComScalingList* sl = s->getScalingList();
  for (UInt sz = 0; sz < 4; sz++)
    for (UInt l = 0; l < 6; l++)
    {
      const Int* c = sl->getScalingListAddress(sz, l);
      const int n = sz == 0 ? 16 : 64;
      for (int i = 0; i < n; i++) out[1 + (sz * 6 + l) * 64 + i] = c[i];
      out[1 + 4 * 6 * 64 + sz * 6 + l] = sl->getScalingListDC(sz, l);
    }
}

/* HM-layout per-CTU metadata.  All arrays [numCTUs][partsPerCTU] in z-scan order (TComDataCU.h:86-157)
 * except sliceIdx [numCTUs].  mv: [numCTUs][parts][2] (hor,ver).  Any pointer may be NULL.            */
void ref_dec_meta(void* h, uint8_t* depth, int8_t* partSize, int8_t* predMode, int8_t* qp, uint8_t* trIdx,
                  uint8_t* cbfY, uint8_t* cbfU, uint8_t* cbfV, uint8_t* tsY, uint8_t* tsU, uint8_t* tsV,
                  int16_t* mv0, int16_t* mv1, int8_t* refIdx0, int8_t* refIdx1,
                  uint8_t* intraDirL, uint8_t* intraDirC, uint8_t* bypass, uint8_t* ipcm, uint8_t* skip, uint8_t* merge,
                  int32_t* sliceIdx)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  const UInt np = p->getNumPartInCU();
  for (UInt a = 0; a < p->getNumCUsInFrame(); a++)
  {
    TComDataCU* cu = p->getCU(a);
    const size_t o = (size_t)a * np;
    if (depth)    memcpy(depth + o, cu->getDepth(), np);
    if (partSize) memcpy(partSize + o, cu->getPartitionSize(), np);
    if (predMode) memcpy(predMode + o, cu->getPredictionMode(), np);
    if (qp)       memcpy(qp + o, cu->getQP(), np);
    if (trIdx)    memcpy(trIdx + o, cu->getTransformIdx(), np);
    if (cbfY)     memcpy(cbfY + o, cu->getCbf(COMPONENT_Y), np);
    if (cbfU)     memcpy(cbfU + o, cu->getCbf(COMPONENT_Cb), np);
    if (cbfV)     memcpy(cbfV + o, cu->getCbf(COMPONENT_Cr), np);
    /* bit 0: m_puhTransformSkip, bits 1-2: m_explicitRdpcmMode (the layout of hmgpu_ctu_meta::transform_skip) */
    {
      uint8_t* ts[3] = { tsY, tsU, tsV };
      for (int c = 0; c < 3; c++)
        if (ts[c])
          for (UInt i = 0; i < np; i++)
          {
            const int mode = cu->getExplicitRdpcmMode(ComponentID(c), i);         /* (NUMBER_OF_RDPCM_MODES = "none parsed": TComDataCU.cpp:435) */
            ts[c][o + i] = (uint8_t)((cu->getTransformSkip(i, ComponentID(c)) ? 1 : 0) | ((mode == RDPCM_HOR || mode == RDPCM_VER ? mode : 0) << 1));
          }
    }
    if (intraDirL) memcpy(intraDirL + o, cu->getIntraDir(CHANNEL_TYPE_LUMA), np);
    if (intraDirC) memcpy(intraDirC + o, cu->getIntraDir(CHANNEL_TYPE_CHROMA), np);
    for (UInt i = 0; i < np; i++)
    {
      if (mv0) { TComMv m = cu->getCUMvField(REF_PIC_LIST_0)->getMv(i); mv0[(o + i) * 2] = m.getHor(); mv0[(o + i) * 2 + 1] = m.getVer(); }
      if (mv1) { TComMv m = cu->getCUMvField(REF_PIC_LIST_1)->getMv(i); mv1[(o + i) * 2] = m.getHor(); mv1[(o + i) * 2 + 1] = m.getVer(); }
      if (refIdx0) refIdx0[o + i] = (int8_t)cu->getCUMvField(REF_PIC_LIST_0)->getRefIdx(i);
      if (refIdx1) refIdx1[o + i] = (int8_t)cu->getCUMvField(REF_PIC_LIST_1)->getRefIdx(i);
      if (bypass) bypass[o + i] = cu->getCUTransquantBypass(i) ? 1 : 0;
      if (ipcm)   ipcm[o + i] = cu->getIPCMFlag(i) ? 1 : 0;
      if (skip)   skip[o + i] = cu->getSkipFlag(i) ? 1 : 0;
      if (merge)  merge[o + i] = cu->getMergeFlag(i) ? 1 : 0;
    }
    if (sliceIdx)
    {
      /* which of the picture's slices the CTU belongs to (slices start at CTU boundaries in HEVC v1) */
      int idx = 0;
      for (UInt s = 0; s < d->top.m_uiSliceIdx; s++)
        if (p->getSlice(s) == cu->getSlice()) idx = s;
      sliceIdx[a] = idx;
    }
  }
}

/* m_crossComponentPredictionAlpha[Cb / Cr] per partition: [numCTUs][parts] */
void ref_dec_ccp_alpha(void* h, int comp, int8_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  const UInt np = p->getNumPartInCU();
  for (UInt a = 0; a < p->getNumCUsInFrame(); a++)
    for (UInt i = 0; i < np; i++) out[(size_t)a * np + i] = (int8_t)p->getCU(a)->getCrossComponentPredictionAlpha(i, ComponentID(comp));
}

/* samples of a CTU in the level / PCM arrays of a component (TComDataCU.cpp:165-173) */
static size_t ctu_elems(TComPic* p, int comp)
{
  return (size_t)(g_uiMaxCUWidth * g_uiMaxCUHeight) >> (p->getComponentScaleX(ComponentID(comp)) + p->getComponentScaleY(ComponentID(comp)));
}

/* coefficient levels as parsed, HM layout (TComDataCU.cpp:165-173): comp 0: [numCTUs][W*H], comp 1,2: [numCTUs][W*H >> (sx + sy)] */
void ref_dec_coeffs(void* h, int comp, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  const size_t n = ctu_elems(p, comp);
  for (UInt a = 0; a < p->getNumCUsInFrame(); a++)
    memcpy(out + a * n, p->getCU(a)->getCoeff(ComponentID(comp)), n * sizeof(TCoeff));
}

/* PCM sample buffers (TComDataCU::getPCMSample: the transmitted samples of PCM CUs; for lossless CUs the decoder parks the
 * reconstruction there, TDecCu::xFillPCMBuffer), same layout as the coefficients.  out[0] of ref_dec_pcm_info: PCM bit depth
 * luma, [1] chroma, [2] pcm enabled, [3] pcm loop filter disabled, [4] transquant bypass enabled */
void ref_dec_pcm(void* h, int comp, int16_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  const size_t n = ctu_elems(p, comp);
  for (UInt a = 0; a < p->getNumCUsInFrame(); a++)
  {
    const Pel* s = p->getCU(a)->getPCMSample(ComponentID(comp));
    for (size_t i = 0; i < n; i++) out[a * n + i] = (int16_t)s[i];
  }
}
void ref_dec_pcm_info(void* h, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComSlice* s = cur(d)->getSlice(0);
  out[0] = s->getSPS()->getPCMBitDepth(CHANNEL_TYPE_LUMA); out[1] = s->getSPS()->getPCMBitDepth(CHANNEL_TYPE_CHROMA);
  out[2] = s->getSPS()->getUsePCM() ? 1 : 0; out[3] = s->getSPS()->getPCMFilterDisableFlag() ? 1 : 0;
  out[4] = s->getPPS()->getTransquantBypassEnableFlag() ? 1 : 0;
}

/* SAO parameters as they stand in TComPicSym (raw as parsed before ref_dec_deblock_sao stage 2,
 * reconstructed after).  out: [numCTUs][3][35] = modeIdc, typeIdc, typeAuxInfo, offset[32]             */
void ref_dec_sao_params(void* h, int32_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  SAOBlkParam* prm = p->getPicSym()->getSAOBlkParam();
  for (UInt a = 0; a < p->getNumCUsInFrame(); a++)
    for (int c = 0; c < 3; c++)
    {
      int32_t* o = out + ((size_t)a * 3 + c) * 35;
      SAOOffset& s = prm[a][c];
      o[0] = (int)s.modeIdc; o[1] = s.typeIdc; o[2] = s.typeAuxInfo;
      for (int k = 0; k < 32; k++) o[3 + k] = s.offset[k];
    }
}

/* visible area of the current picture's reconstruction planes, dense (stride = width) */
static void copy_planes(TComPicYuv* yuv, int16_t* y, int16_t* cb, int16_t* cr)
{
  int16_t* dst[3] = { y, cb, cr };
  for (int c = 0; c < 3; c++)
  {
    if (!dst[c]) continue;
    const ComponentID id = ComponentID(c);
    const int w = yuv->getWidth(id), hh = yuv->getHeight(id), st = yuv->getStride(id);
    const Pel* src = yuv->getAddr(id);
    for (int r = 0; r < hh; r++) memcpy(dst[c] + (size_t)r * w, src + (size_t)r * st, w * sizeof(Pel));
  }
}

void ref_dec_planes(void* h, int16_t* y, int16_t* cb, int16_t* cr)
{
  RefDec* d = (RefDec*)h;
  copy_planes(cur(d)->getPicYuvRec(), y, cb, cr);
}

/* planes of a picture in the DPB by POC (reference pictures); returns 0 if found */
int ref_dec_dpb_planes(void* h, int poc, int16_t* y, int16_t* cb, int16_t* cr)
{
  RefDec* d = (RefDec*)h;
  for (TComList<TComPic*>::iterator it = d->top.m_cListPic.begin(); it != d->top.m_cListPic.end(); ++it)
    if ((*it) != cur(d) && (*it)->getReconMark() && (*it)->getPOC() == poc)
    {
      copy_planes((*it)->getPicYuvRec(), y, cb, cr);
      return 0;
    }
  return -1;
}

/* advance the filter stage of the pending picture by one step:
 *   stage 0 -> 1 : TComLoopFilter::loopFilterPic          (TDecGop.cpp:165-167)
 *   stage 1 -> 2 : reconstructBlkSAOParams + SAOProcess + PCMLFDisableProcess (TDecGop.cpp:169-174)
 * returns the new stage */
int ref_dec_filter_step(void* h)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  TComSlice* s = p->getSlice(p->getCurrSliceIdx());
  if (d->stage == 0)
  {
    d->top.m_cLoopFilter.setCfg(s->getPPS()->getLoopFilterAcrossTilesEnabledFlag());
    d->top.m_cLoopFilter.loopFilterPic(p);
    d->stage = 1;
  }
  else if (d->stage == 1)
  {
    if (s->getSPS()->getUseSAO())
    {
      d->top.m_cSAO.reconstructBlkSAOParams(p, p->getPicSym()->getSAOBlkParam());
      d->top.m_cSAO.SAOProcess(p);
      d->top.m_cSAO.PCMLFDisableProcess(p);
    }
    d->stage = 2;
  }
  return d->stage;
}

/* finish the picture: remaining steps of filterPicture (compressMotion, hash check, marks) and of
 * executeLoopFilters (list sort, CU decoder teardown).  Returns 1 if the SEI hash matched or none,
 * 0 on mismatch.  md5[16] receives the MD5 of plane `comp` bytes the way TComPicYuvMD5 computes it
 * (calcMD5, TComPicYuvMD5.cpp:183): 48 bytes = 3 planes x 16.                                        */
/* the two other picture hashes of the decoded-picture-hash SEI (TComPicYuvMD5.cpp:89-170) on the finished picture:
 * out[0..5] CRC (2 bytes per component), out[6..17] checksum (4 bytes per component).  Call after both filter steps. */
void ref_dec_hashes(void* h, uint8_t* out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  TComDigest dig;
  calcCRC(*p->getPicYuvRec(), dig);
  for (size_t i = 0; i < 6 && i < dig.hash.size(); i++) out[i] = dig.hash[i];
  calcChecksum(*p->getPicYuvRec(), dig);
  for (size_t i = 0; i < 12 && i < dig.hash.size(); i++) out[6 + i] = dig.hash[i];
}

int ref_dec_finish(void* h, uint8_t* md5out)
{
  RefDec* d = (RefDec*)h; TComPic* p = cur(d);
  while (d->stage < 2) ref_dec_filter_step(h);
  p->compressMotion();
  if (md5out)
  {
    TComDigest dig;
    calcMD5(*p->getPicYuvRec(), dig);
    for (size_t i = 0; i < dig.hash.size() && i < 48; i++) md5out[i] = dig.hash[i];
  }
  int ok = 1;
  {
    SEIMessages pictureHashes = getSeisByType(p->getSEIs(), SEI::DECODED_PICTURE_HASH);
    if (pictureHashes.size() > 0)
    {
      const SEIDecodedPictureHash* hash = (SEIDecodedPictureHash*)*(pictureHashes.begin());
      if (hash->method == SEIDecodedPictureHash::MD5)
      {
        TComDigest dig; calcMD5(*p->getPicYuvRec(), dig);
        if (dig != hash->m_digest) { ok = 0; d->numMismatch++; }
      }
    }
  }
  p->setOutputMark(false);          /* "output" happens right here: the harness reads planes itself */
  p->setReconMark(true);
  TComSlice::sortPicList(d->top.m_cListPic);
  d->top.m_cCuDecoder.destroy();
  d->top.m_bFirstSliceInPicture = true;
  d->pending = false;
  return ok;
}

/* time HM's own reconstruction+filter path on this stream: decodes everything, returns the seconds HM's
 * TDecGop accumulates around decompressSlice+filterPicture is not reachable without stdout parsing, so
 * the harness simply clocks ref_dec_next/ref_dec_finish from the caller side (see bench.py).            */

} /* extern "C" */
