// facade.cpp -- the libHMDecoder-compatible C interface (include/hmdec.h) on top of hmdec::Decoder.
// Protocol and output rules follow ChristianFeldmann/libHM source/App/libHMDecoder/libHMDecoder.cpp:112-339 (push / get_picture)
// and :450-720 (internals), re-stated on this decoder's own data structures.
#include "../../include/hmdec.h"

#include <cstdlib>
#include <cstring>
#include <string>

#include "decoder.h"

using namespace hmdec;

namespace {
struct Wrapper {
  Decoder dec;
  int max_temporal_layer = -1;
  bool flush_output = false, schedule_flush = false, loop_filtered = false, have_list = false;
  std::vector<libHMDec_BlockValue> internals;
};
int g_bit_depth[2] = {8, 8};           // HM keeps the bit depths in globals (g_bitDepth); the getter has no context argument

PicData* as_pic(libHMDec_picture* p) { return static_cast<PicData*>(p); }
}  // namespace

extern "C" {

const char* libHMDec_get_version(void) { return "16.0"; }

libHMDec_context* libHMDec_new_decoder(void) {
  try {
    Wrapper* w = new Wrapper();
    if (const char* e = getenv("HMDEC_THREADS")) w->dec.set_threads(atoi(e));      // lets an unmodified libHM client use parser threads
    return w;
  } catch (...) { return nullptr; }
}

libHMDec_error libHMDec_free_decoder(libHMDec_context* ctx) {
  if (!ctx) return LIBHMDEC_ERROR;
  delete static_cast<Wrapper*>(ctx);
  return LIBHMDEC_OK;
}

void libHMDec_set_SEI_Check(libHMDec_context* ctx, bool check_hash) {
  if (ctx) static_cast<Wrapper*>(ctx)->dec.set_check_hash(check_hash);
}

void libHMDec_set_max_temporal_layer(libHMDec_context* ctx, int max_layer) {
  if (ctx) static_cast<Wrapper*>(ctx)->max_temporal_layer = max_layer;
}

libHMDec_error libHMDec_push_nal_unit(libHMDec_context* ctx, const void* data8, int length, bool eof, bool& bNewPicture, bool& checkOutputPictures) {
  Wrapper* w = static_cast<Wrapper*>(ctx);
  bNewPicture = false;
  checkOutputPictures = false;
  if (!w) return LIBHMDEC_ERROR;
  if (length <= 0 || !data8) return LIBHMDEC_ERROR_READ_ERROR;
  if (length < 4 && !eof) return LIBHMDEC_ERROR_READ_ERROR;
  int nal_type = -1;
  try {
    bNewPicture = w->dec.push(static_cast<const uint8_t*>(data8), (size_t)length, w->max_temporal_layer, &nal_type);
    if (eof || nal_type == NAL_EOS) {
      if (!w->loop_filtered || !eof) w->dec.finish_picture();
      w->loop_filtered = nal_type == NAL_EOS;
    } else if (bNewPicture) {
      w->loop_filtered = false;
    }
  } catch (const Unsupported& e) {
    w->dec.set_error(std::string("unsupported: ") + e.what());
    return LIBHMDEC_ERROR;
  } catch (const ParseError& e) {
    w->dec.set_error(std::string("bitstream: ") + e.what());
    return LIBHMDEC_ERROR_READ_ERROR;
  } catch (const std::exception& e) {
    w->dec.set_error(e.what());
    return LIBHMDEC_ERROR;
  }
  if (const Sps* sps = w->dec.active_sps()) { g_bit_depth[0] = sps->bit_depth_luma; g_bit_depth[1] = sps->bit_depth_chroma; }
  if (w->dec.threaded()) {
    // parser threads: pictures reach the output queue as they leave the pipeline (in decoding order, with the same output rule);
    // the points where libHM flushes -- a new IDR / BLA picture, the end of the stream -- first wait for the pipeline to drain
    try {
      const bool irap = nal_type == NAL_IDR_W_RADL || nal_type == NAL_IDR_N_LP || nal_type == NAL_BLA_N_LP || nal_type == NAL_BLA_W_RADL || nal_type == NAL_BLA_W_LP;
      if ((bNewPicture && irap) || eof || nal_type == NAL_EOS) w->dec.queue_flush();
    } catch (const std::exception& e) {
      w->dec.set_error(e.what());
      return LIBHMDEC_ERROR_READ_ERROR;
    }
    checkOutputPictures = true;
    if (eof) {                             // a picture dropped while the pipeline drained: nothing follows that could report it
      const std::string late = w->dec.take_deferred_error();
      if (!late.empty()) { w->dec.set_error(std::string("bitstream: ") + late); return LIBHMDEC_ERROR_READ_ERROR; }
    }
    return LIBHMDEC_OK;
  }
  w->flush_output = false;
  const bool irap_flush = nal_type == NAL_IDR_W_RADL || nal_type == NAL_IDR_N_LP || nal_type == NAL_BLA_N_LP || nal_type == NAL_BLA_W_RADL || nal_type == NAL_BLA_W_LP;
  if (bNewPicture && irap_flush) { checkOutputPictures = true; w->flush_output = true; }
  if (nal_type == NAL_EOS) checkOutputPictures = true;
  const bool vcl = nal_type >= NAL_TRAIL_N && nal_type <= NAL_RSV_VCL31;
  if ((bNewPicture || (!bNewPicture && vcl)) && w->dec.pictures_decoded() > 0) checkOutputPictures = true;
  if (eof) { checkOutputPictures = true; w->schedule_flush = true; }
  if (checkOutputPictures) w->dec.begin_output_scan(w->max_temporal_layer);
  return LIBHMDEC_OK;
}

libHMDec_picture* libHMDec_get_picture(libHMDec_context* ctx) {
  Wrapper* w = static_cast<Wrapper*>(ctx);
  if (!w) return nullptr;
  if (w->dec.threaded()) {
    PicData* q = w->dec.pop_output();
    if (q) w->dec.fetch_planes(q);
    return q;
  }
  PicData* p = w->dec.next_output(w->flush_output);
  if (!p) {
    if (w->flush_output) { w->dec.last_display_poc = -(1 << 30); w->flush_output = false; }
    if (w->schedule_flush) {
      w->flush_output = true;
      w->schedule_flush = false;
      w->dec.begin_output_scan(w->max_temporal_layer);
      return libHMDec_get_picture(ctx);
    }
    return nullptr;
  }
  w->dec.fetch_planes(p);              // the samples leave the device when the application asks for the picture
  return p;
}

int libHMDEC_get_POC(libHMDec_picture* pic) { return pic ? as_pic(pic)->poc : -1; }

int libHMDEC_get_picture_width(libHMDec_picture* pic, libHMDec_ColorComponent c) {
  if (!pic || c < LIBHMDEC_LUMA || c > LIBHMDEC_CHROMA_V) return -1;
  return as_pic(pic)->width >> (c == LIBHMDEC_LUMA ? 0 : as_pic(pic)->csx);
}
int libHMDEC_get_picture_height(libHMDec_picture* pic, libHMDec_ColorComponent c) {
  if (!pic || c < LIBHMDEC_LUMA || c > LIBHMDEC_CHROMA_V) return -1;
  return as_pic(pic)->height >> (c == LIBHMDEC_LUMA ? 0 : as_pic(pic)->csy);
}
int libHMDEC_get_picture_stride(libHMDec_picture* pic, libHMDec_ColorComponent c) { return libHMDEC_get_picture_width(pic, c); }

short* libHMDEC_get_image_plane(libHMDec_picture* pic, libHMDec_ColorComponent c) {
  if (!pic || c < LIBHMDEC_LUMA || c > LIBHMDEC_CHROMA_V) return nullptr;
  PicData* p = as_pic(pic);
  if (!p->planes_valid) return nullptr;
  if (c != LIBHMDEC_LUMA && p->num_comps == 1) return nullptr;        // monochrome: HM allocates no chroma buffers (TComPicYuv::create)
  return p->plane[c].data();
}

libHMDec_ChromaFormat libHMDEC_get_chroma_format(libHMDec_picture* pic) {
  if (!pic) return LIBHMDEC_CHROMA_UNKNOWN;
  switch (as_pic(pic)->chroma_format) {
    case 0: return LIBHMDEC_CHROMA_400;
    case 2: return LIBHMDEC_CHROMA_422;
    case 3: return LIBHMDEC_CHROMA_444;
    default: return LIBHMDEC_CHROMA_420;
  }
}

int libHMDEC_get_internal_bit_depth(libHMDec_ColorComponent c) {
  if (c == LIBHMDEC_LUMA) return g_bit_depth[0];
  if (c == LIBHMDEC_CHROMA_U || c == LIBHMDEC_CHROMA_V) return g_bit_depth[1];
  return -1;
}

// ---- internals: walk the CU / PU / TU structure of a picture out of the per-partition arrays
static void tu_values(Wrapper* w, const PicData& p, int x, int y, int log2, int tr_depth, libHMDec_info_type type) {
  const size_t part = p.part_at(x, y);
  if (tr_depth < p.tr_idx[part]) {
    const int h = 1 << (log2 - 1);
    for (int i = 0; i < 4; i++) tu_values(w, p, x + (i & 1) * h, y + (i >> 1) * h, log2 - 1, tr_depth + 1, type);
  }
  libHMDec_BlockValue b;
  memset(&b, 0, sizeof(b));
  b.x = (unsigned short)x; b.y = (unsigned short)y; b.w = b.h = (unsigned short)(1 << log2);
  switch (type) {
    case LIBHMDEC_TU_CBF_Y: b.value = (p.cbf[0][part] >> tr_depth) & 1; break;
    case LIBHMDEC_TU_CBF_CB: b.value = (p.cbf[1][part] >> tr_depth) & 1; break;
    case LIBHMDEC_TU_CBF_CR: b.value = (p.cbf[2][part] >> tr_depth) & 1; break;
    case LIBHMDEC_TU_COEFF_TR_SKIP_Y: b.value = p.ts[0][part] != 0; break;
    case LIBHMDEC_TU_COEFF_TR_SKIP_Cb: b.value = p.ts[1][part] != 0; break;
    case LIBHMDEC_TU_COEFF_TR_SKIP_Cr: b.value = p.ts[2][part] != 0; break;
    default: {
      const int c = type == LIBHMDEC_TU_COEFF_ENERGY_Y ? 0 : type == LIBHMDEC_TU_COEFF_ENERGY_CB ? 1 : 2;
      const size_t ctb = p.ctb_at(x, y), z = part - ctb * p.parts;
      const int16_t* lv = p.level_src(c, ctb, z);
      const int n = c == 0 ? (1 << (2 * log2)) : (1 << (2 * log2 - 2));
      int64_t e = 0;
      for (int i = 0; lv && i < n; i++) e += (int64_t)lv[i] * lv[i];
      b.value = e > 0x7fffffff ? 0x7fffffff : (int)e;
      break;
    }
  }
  w->internals.push_back(b);
}

static void cu_values(Wrapper* w, const PicData& p, int x, int y, int log2, int depth, libHMDec_info_type type) {
  const int size = 1 << log2;
  if (x >= p.width || y >= p.height) return;
  const size_t part = p.part_at(x, y);
  const bool boundary = x + size > p.width || y + size > p.height;
  if (boundary || p.depth[part] > depth) {
    if (log2 <= 3) return;
    const int h = size >> 1;
    for (int i = 0; i < 4; i++) cu_values(w, p, x + (i & 1) * h, y + (i >> 1) * h, log2 - 1, depth + 1, type);
    return;
  }
  if (p.part_size[part] == HMGPU_SIZE_NONE) return;        // never decoded
  const bool intra = p.pred_mode[part] == HMGPU_MODE_INTRA, inter = p.pred_mode[part] == HMGPU_MODE_INTER;
  libHMDec_BlockValue b;
  memset(&b, 0, sizeof(b));
  b.x = (unsigned short)x; b.y = (unsigned short)y; b.w = b.h = (unsigned short)size;
  switch (type) {
    case LIBHMDEC_CU_PREDICTION_MODE: b.value = p.pred_mode[part]; w->internals.push_back(b); return;
    case LIBHMDEC_CU_TRQ_BYPASS: if (p.has_bypass) { b.value = p.bypass[part]; w->internals.push_back(b); } return;
    case LIBHMDEC_CU_SKIP_FLAG: b.value = p.skip[part]; w->internals.push_back(b); return;
    case LIBHMDEC_CU_PART_MODE: b.value = p.part_size[part]; w->internals.push_back(b); return;
    case LIBHMDEC_CU_INTRA_MODE_LUMA: if (intra) { b.value = p.intra_dir[0][part]; w->internals.push_back(b); } return;
    case LIBHMDEC_CU_INTRA_MODE_CHROMA: if (intra) { b.value = p.intra_dir[1][part]; w->internals.push_back(b); } return;
    case LIBHMDEC_CU_ROOT_CBF: if (!inter) { b.value = ((p.cbf[0][part] | p.cbf[1][part] | p.cbf[2][part]) & 1); w->internals.push_back(b); } return;
    default: break;
  }
  if (type >= LIBHMDEC_PU_MERGE_FLAG && type <= LIBHMDEC_PU_MV_1) {
    if (!inter) return;
    const int ps = p.part_size[part], h = size >> 1, q = size >> 2;
    int n = ps == HMGPU_SIZE_2Nx2N ? 1 : ps == HMGPU_SIZE_NxN ? 4 : 2;
    int rx[4] = {x, x, x, x}, ry[4] = {y, y, y, y}, rw[4] = {size, size, size, size}, rh[4] = {size, size, size, size};
    switch (ps) {
      case HMGPU_SIZE_2NxN: rh[0] = rh[1] = h; ry[1] = y + h; break;
      case HMGPU_SIZE_Nx2N: rw[0] = rw[1] = h; rx[1] = x + h; break;
      case HMGPU_SIZE_NxN: for (int i = 0; i < 4; i++) { rw[i] = rh[i] = h; rx[i] = x + (i & 1) * h; ry[i] = y + (i >> 1) * h; } break;
      case HMGPU_SIZE_2NxnU: rh[0] = q; rh[1] = size - q; ry[1] = y + q; break;
      case HMGPU_SIZE_2NxnD: rh[0] = size - q; rh[1] = q; ry[1] = y + size - q; break;
      case HMGPU_SIZE_nLx2N: rw[0] = q; rw[1] = size - q; rx[1] = x + q; break;
      case HMGPU_SIZE_nRx2N: rw[0] = size - q; rw[1] = q; rx[1] = x + size - q; break;
      default: break;
    }
    for (int i = 0; i < n; i++) {
      const size_t pp = p.part_at(rx[i], ry[i]);
      libHMDec_BlockValue v;
      memset(&v, 0, sizeof(v));
      v.x = (unsigned short)rx[i]; v.y = (unsigned short)ry[i]; v.w = (unsigned short)rw[i]; v.h = (unsigned short)rh[i];
      const int dir = p.inter_dir[pp];     // HM's interDir: 1 = list 0, 2 = list 1, 3 = both
      switch (type) {
        case LIBHMDEC_PU_MERGE_FLAG: v.value = p.merge[pp]; break;
        case LIBHMDEC_PU_MERGE_INDEX: if (p.merge[pp]) v.value = p.merge_idx[pp]; break;
        case LIBHMDEC_PU_UNI_BI_PREDICTION: v.value = dir; break;
        case LIBHMDEC_PU_REFERENCE_POC_0: v.value = p.ref_idx[0][pp]; break;
        case LIBHMDEC_PU_MV_0: v.value = p.mv[0][2 * pp]; v.value2 = p.mv[0][2 * pp + 1]; break;
        case LIBHMDEC_PU_REFERENCE_POC_1: if (dir == 2) v.value = p.ref_idx[1][pp]; break;
        case LIBHMDEC_PU_MV_1: if (dir == 2) { v.value = p.mv[1][2 * pp]; v.value2 = p.mv[1][2 * pp + 1]; } break;
        default: break;
      }
      w->internals.push_back(v);
    }
    return;
  }
  tu_values(w, p, x, y, log2, 0, type);
}

std::vector<libHMDec_BlockValue>* libHMDEC_get_internal_info(libHMDec_context* ctx, libHMDec_picture* pic, libHMDec_info_type type) {
  Wrapper* w = static_cast<Wrapper*>(ctx);
  if (!w) return nullptr;
  w->internals.clear();
  if (!pic) return nullptr;
  const PicData& p = *as_pic(pic);
  const int ctb = 1 << p.log2_ctb;
  for (int rs = 0; rs < p.num_ctbs; rs++) {
    const int x = (rs % p.ctbs_w) * ctb, y = (rs / p.ctbs_w) * ctb;
    if (type == LIBHMDEC_CTU_SLICE_INDEX) {
      libHMDec_BlockValue b;
      memset(&b, 0, sizeof(b));
      b.x = (unsigned short)x; b.y = (unsigned short)y; b.w = b.h = (unsigned short)ctb;
      b.value = p.slice_idx[rs];
      w->internals.push_back(b);
    } else {
      cu_values(w, p, x, y, p.log2_ctb, 0, type);
    }
  }
  return &w->internals;
}

libHMDec_error libHMDEC_clear_internal_info(libHMDec_context* ctx) {
  Wrapper* w = static_cast<Wrapper*>(ctx);
  if (!w) return LIBHMDEC_ERROR;
  w->internals.clear();
  w->internals.shrink_to_fit();
  return LIBHMDEC_OK;
}

// ------------------------------------------------------------------------------------------------ this library's own additions
void hmdec_set_device(libHMDec_context* ctx, int ordinal) { if (ctx) static_cast<Wrapper*>(ctx)->dec.set_device(ordinal); }
void hmdec_set_devices(libHMDec_context* ctx, const int* ordinals, int n) { if (ctx && ordinals && n > 0 && n <= 32) static_cast<Wrapper*>(ctx)->dec.set_devices(ordinals, n); }
int hmdec_num_devices(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.num_devices() : -1; }
unsigned long long hmdec_transfer_bytes(libHMDec_context* ctx) { return ctx ? (unsigned long long)static_cast<Wrapper*>(ctx)->dec.transfer_bytes() : 0ull; }
void hmdec_set_threads(libHMDec_context* ctx, int n) { if (ctx) static_cast<Wrapper*>(ctx)->dec.set_threads(n); }
void hmdec_set_parse_only(libHMDec_context* ctx, int on) { if (ctx) static_cast<Wrapper*>(ctx)->dec.set_parse_only(on != 0); }
int hmdec_hash_mismatches(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.hash_mismatches() : -1; }
int hmdec_pictures_decoded(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.pictures_decoded() : -1; }
void hmdec_set_device_md5(libHMDec_context* ctx, int on) { if (ctx) static_cast<Wrapper*>(ctx)->dec.set_device_md5(on != 0); }
int hmdec_device_batches(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.device_batches() : -1; }
const char* hmdec_last_error(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.last_error().c_str() : ""; }
libHMDec_picture* hmdec_last_decoded_picture(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.last_decoded() : nullptr; }

libHMDec_picture* hmdec_open_picture(libHMDec_context* ctx) { return ctx ? static_cast<Wrapper*>(ctx)->dec.open_picture() : nullptr; }

int hmdec_picture_array(libHMDec_picture* pic, const char* name, const void** data, int64_t* bytes) {
  if (!pic || !name || !data || !bytes) return 1;
  PicData& p = *as_pic(pic);
  const std::string n(name);
  auto give = [&](const void* d, size_t b) { *data = d; *bytes = (int64_t)b; return 0; };
  auto idx = [&](size_t prefix) { return n.size() == prefix + 1 ? n[prefix] - '0' : -1; };
  if (n == "depth") return give(p.depth.data(), p.depth.size());
  if (n == "part_size") return give(p.part_size.data(), p.part_size.size());
  if (n == "pred_mode") return give(p.pred_mode.data(), p.pred_mode.size());
  if (n == "qp") return give(p.qp.data(), p.qp.size());
  if (n == "tr_idx") return give(p.tr_idx.data(), p.tr_idx.size());
  if (n == "bypass") return give(p.bypass.data(), p.bypass.size());
  if (n == "ipcm") return give(p.ipcm.data(), p.ipcm.size());
  if (n == "skip") return give(p.skip.data(), p.skip.size());
  if (n == "merge") return give(p.merge.data(), p.merge.size());
  if (n == "slice_idx") return give(p.slice_idx.data(), p.slice_idx.size() * 2);
  if (n == "tile_idx") return give(p.tile_idx.data(), p.tile_idx.size() * 2);
  if (n == "sao") return give(p.sao.data(), p.sao.size() * sizeof(hmgpu_sao_param));
  if (n.compare(0, 3, "cbf") == 0 && idx(3) >= 0 && idx(3) < 3) return give(p.cbf[idx(3)].data(), p.cbf[idx(3)].size());
  if (n.compare(0, 2, "ts") == 0 && idx(2) >= 0 && idx(2) < 3) return give(p.ts[idx(2)].data(), p.ts[idx(2)].size());
  if (n.compare(0, 2, "mv") == 0 && idx(2) >= 0 && idx(2) < 2) return give(p.mv[idx(2)].data(), p.mv[idx(2)].size() * 2);
  if (n.compare(0, 7, "ref_idx") == 0 && idx(7) >= 0 && idx(7) < 2) return give(p.ref_idx[idx(7)].data(), p.ref_idx[idx(7)].size());
  if (n.compare(0, 9, "intra_dir") == 0 && idx(9) >= 0 && idx(9) < 2) return give(p.intra_dir[idx(9)].data(), p.intra_dir[idx(9)].size());
  if (n.compare(0, 5, "coeff") == 0 && idx(5) >= 0 && idx(5) < 3) return give(p.coeff[idx(5)].data(), p.coeff[idx(5)].size() * 2);
  if (n.compare(0, 3, "pcm") == 0 && idx(3) >= 0 && idx(3) < 3) return give(p.pcm[idx(3)].data(), p.pcm[idx(3)].size() * 2);
  if (n.compare(0, 3, "ccp") == 0 && idx(3) >= 0 && idx(3) < 2 && !p.ccp[idx(3)].empty()) return give(p.ccp[idx(3)].data(), p.ccp[idx(3)].size());
  if (n.compare(0, 5, "plane") == 0 && idx(5) >= 0 && idx(5) < 3 && p.planes_valid) return give(p.plane[idx(5)].data(), p.plane[idx(5)].size() * 2);
  return 1;
}

int hmdec_picture_num_slices(libHMDec_picture* pic) { return pic ? (int)as_pic(pic)->slices.size() : -1; }

int hmdec_picture_slice_params(libHMDec_picture* pic, int slice, void* out, void* lists_out) {
  if (!pic || !out) return 1;
  PicData& p = *as_pic(pic);
  if (slice < 0 || slice >= (int)p.slices.size()) return 1;
  memcpy(out, &p.slices[slice]->params, sizeof(hmgpu_slice_params));
  if (lists_out && p.slices[slice]->scaling_lists) memcpy(lists_out, p.slices[slice]->scaling_lists.get(), sizeof(hmgpu_scaling_lists));
  return 0;
}

int hmdec_picture_geometry(libHMDec_picture* pic, int32_t out[12]) {
  if (!pic || !out) return 1;
  const PicData& p = *as_pic(pic);
  const int32_t v[12] = {p.width, p.height, p.log2_ctb, p.bit_depth[0], p.bit_depth[1], p.pcm_bit_depth[0], p.pcm_bit_depth[1],
                         p.pcm_lf_disable, p.strong_intra, p.sao_enabled, p.lf_across_tiles, p.num_ctbs};
  memcpy(out, v, sizeof(v));
  return 0;
}

int hmdec_picture_range_ext_flags(libHMDec_picture* pic) { return pic ? as_pic(pic)->range_ext_flags : 0; }
int hmdec_picture_chroma_format(libHMDec_picture* pic) { return pic ? as_pic(pic)->chroma_format : -1; }
int hmdec_picture_sao_offset_shift(libHMDec_picture* pic, int chroma) { return pic ? as_pic(pic)->sao_offset_shift[chroma ? 1 : 0] : 0; }

int hmdec_picture_conformance_window(libHMDec_picture* pic, int32_t window[4]) {
  if (!pic || !window) return 1;
  for (int i = 0; i < 4; i++) window[i] = as_pic(pic)->conf_window[i];
  return 0;
}

int hmdec_internal_info(libHMDec_context* ctx, libHMDec_picture* pic, int type, const libHMDec_BlockValue** data) {
  if (!data || type < LIBHMDEC_CTU_SLICE_INDEX || type > LIBHMDEC_TU_COEFF_ENERGY_CR) return -1;
  std::vector<libHMDec_BlockValue>* v = libHMDEC_get_internal_info(ctx, pic, (libHMDec_info_type)type);
  if (!v) return -1;
  *data = v->data();
  return (int)v->size();
}

int hmdec_picture_hash_sei(libHMDec_picture* pic, uint8_t digest[48]) {
  if (!pic) return -1;
  PicData& p = *as_pic(pic);
  if (digest) memcpy(digest, p.sei_hash, 48);
  return p.sei_hash_method;
}

}  // extern "C"
