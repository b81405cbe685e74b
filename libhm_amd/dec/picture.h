// picture.h -- what the parser leaves behind for one picture: HM's per-CTU TComDataCU arrays (4x4 partitions in z-scan order inside
// each CTU, TComDataCU.h:86-157), coefficient levels in HM's TU layout, SAO parameters as parsed, and the slice table.  These are
// exactly the arrays hmgpu_decompress_slice / hmgpu_filter_picture take (include/hmgpu.h), so nothing is converted on the way.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/hmgpu.h"
#include "params.h"

namespace hmdec {

// The arrays the device reads (and the planes it writes) live in page-locked memory when a device is in use: hmgpu stages them with
// asynchronous DMA while the parser goes on with the next picture.  The switch is per thread and read at allocation time.
void* host_alloc(size_t bytes);
void host_free(void* p);
void host_alloc_use_pinned(bool on);
template <class T> struct HostAlloc {
  using value_type = T;
  HostAlloc() = default;
  template <class U> HostAlloc(const HostAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(host_alloc(n * sizeof(T))); }
  void deallocate(T* p, size_t) { host_free(p); }
  template <class U> bool operator==(const HostAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const HostAlloc<U>&) const { return false; }
};
template <class T> using HostVec = std::vector<T, HostAlloc<T>>;

// An array of the parser that the device reads: either storage of its own (no device: parse-only runs) or a view into the picture's
// staging block (hmgpu_staging_alloc: ONE page-locked block that mirrors the device's array layout, moved in one DMA).  The subset
// of std::vector the parser uses.
template <class T> struct Arr {
  T* p = nullptr;
  size_t n = 0;
  HostVec<T> own;
  void assign(size_t count, T v) { own.assign(count, v); p = own.data(); n = count; }
  void bind(const T* ptr, size_t count) { own.clear(); own.shrink_to_fit(); p = const_cast<T*>(ptr); n = count; }
  void detach() { if (p && own.empty() && n) { own.assign(p, p + n); p = own.data(); } }      // a view becomes a copy of its own
  bool is_view() const { return p && own.empty() && n; }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  T* begin() { return p; }
  T* end() { return p + n; }
  const T* begin() const { return p; }
  const T* end() const { return p + n; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

struct ZScan {                       // raster (4x4 units inside a CTU) <-> HM z-scan order (g_auiRasterToZscan / g_auiZscanToRaster)
  int log2_ctb = 0, n4 = 0;          // n4: 4x4 units per CTU side
  std::vector<uint16_t> r2z, z2r;
  void init(int log2_ctb_size) {
    log2_ctb = log2_ctb_size;
    n4 = 1 << (log2_ctb - 2);
    r2z.assign(n4 * n4, 0);
    z2r.assign(n4 * n4, 0);
    for (int y = 0; y < n4; y++)
      for (int x = 0; x < n4; x++) {
        int z = 0;
        for (int b = 0; b < log2_ctb - 2; b++) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
        r2z[y * n4 + x] = (uint16_t)z;
        z2r[z] = (uint16_t)(y * n4 + x);
      }
  }
};

struct PicData;
struct SliceInfo {                   // one slice (independent segment + its dependent segments)
  PicData* ref_pics[2][16];
  hmgpu_slice_params params;
  std::unique_ptr<hmgpu_scaling_lists> scaling_lists;
  int type = SLICE_I;
  int ref_poc[2][16];                // POC and marking of every reference at the time the slice was decoded (TMVP of later pictures)
  bool ref_is_lt[2][16];
  int first_ctb_ts = 0, num_ctbs = 0;
  SliceHeader header;
};

struct PicData {
  // geometry
  int width = 0, height = 0, log2_ctb = 0, ctbs_w = 0, ctbs_h = 0, num_ctbs = 0, parts = 0;
  const ZScan* zs = nullptr;
  // per partition, [num_ctbs][parts]
  Arr<uint8_t> depth, tr_idx, cbf[3], ts[3], intra_dir[2], bypass, ipcm;
  HostVec<uint8_t> skip, merge, merge_idx, inter_dir;             // (parser-only: merge / AMVP derivation of later CUs and pictures)
  Arr<int8_t> part_size, pred_mode, qp, ref_idx[2];
  Arr<int16_t> mv[2];            // {hor, ver}
  // per CTB
  Arr<uint16_t> slice_idx, tile_idx;
  std::vector<int32_t> slice_addr;   // SliceAddrRs of the slice that decoded the CTB, -1 = not (yet) decoded
  // levels: HM's dense TU layout (coefficients of the TU at partition z of a CTU at 16 z / 4 z of the CTU's share), or COMPACT
  // (hmgpu_coeffs::ctu_level_start): only the coded TUs, one after the other in parsing order = z order per component, with the offset
  // of every CTU's first one -- written that way when the picture is parsed by one thread from front to back (no wavefront / tile
  // workers) and handed to the device (the transfer shrinks to the coded part)
  Arr<int16_t> coeff[3];
  Arr<uint32_t> level_start[3];      // compact: [num_ctbs + 1]
  bool compact = false;
  uint32_t level_cursor[3] = {0, 0, 0};
  std::vector<uint32_t> tu_off[3];   // compact: offset of the TU that starts at a partition (what the dense layout gives by arithmetic)
  HostVec<int16_t> pcm[3];           // PCM samples in HM's dense layout
  HostVec<int8_t> ccp[2];            // cross_component_prediction weights of Cb / Cr per partition (4:4:4 with the PPS flag; else empty)
  int csx = 1, csy = 1, cshift = 2;  // chroma subsampling; a chroma block has (luma samples >> cshift) samples
  // the staging block the arrays above are views of (null: they own their storage)
  hmgpu_ctx* stg_ctx = nullptr;
  hmgpu_staging* stg = nullptr;
  hmgpu_ctu_meta stg_meta;
  hmgpu_coeffs stg_co;
  HostVec<hmgpu_sao_param> sao;      // [num_ctbs][3]
  std::vector<std::unique_ptr<SliceInfo>> slices;
  // picture state (8.3)
  int poc = 0, nal_type = 0, temporal_id = 0, conf_window[4] = {0, 0, 0, 0};
  bool is_reference = false, is_long_term = false, needed_for_output = false, pic_output = true, decoded = false, filtered = false;
  bool has_pcm = false, has_bypass = false, lent = false;     // lent: handed to the application by the last output scan
  // frame-parallel parsing (decoder.cpp): progress of the parser thread that owns the picture, and who still reads its arrays
  std::atomic<int> rows_done{0};     // CTB rows whose motion data is final (temporal MV prediction of later pictures waits on it)
  std::atomic<bool> parse_done{true};
  std::atomic<int> users{0};         // pictures in flight that predict from this one
  bool in_flight = false;
  bool sao_enabled = false, lf_across_tiles = true;   // of the parameter sets the picture was decoded with
  int bit_depth[2] = {8, 8}, pcm_bit_depth[2] = {8, 8};
  bool pcm_lf_disable = false, strong_intra = false;
  int sao_offset_shift[2] = {0, 0};            // PPS log2_sao_offset_scale_{luma,chroma}
  int num_comps = 3;                           // 1: monochrome (chroma_format_idc 0)
  int chroma_format = 1;                       // chroma_format_idc of the active SPS
  int range_ext_flags = 0;                     // HMGPU_REXT_* of the active SPS
  hmgpu_pic handle = HMGPU_NO_PIC;   // the same handle in every device context of the decoder
  int home = 0;                      // the context that decoded the picture (Decoder::gpus_) ...
  uint32_t present = 0;              // ... and the contexts that hold its finished samples (bit k: gpus_[k])
  uint64_t submit_seq = 0;           // device submission that last read these arrays
  // output side
  bool planes_valid = false;
  std::atomic<uint64_t> dl_ticket{0};   // a download of the planes is under way (hmgpu_picture_download_begin): planes_valid once it has been waited for
  HostVec<int16_t> plane[3];
  uint8_t sei_hash[3][16];
  int sei_hash_method = 0;           // 0 = none, 1 = MD5, 2 = CRC, 3 = checksum
  std::atomic<bool> hash_mismatch{false};

  PicData() { memset(&stg_meta, 0, sizeof(stg_meta)); memset(&stg_co, 0, sizeof(stg_co)); }
  PicData(const PicData&) = delete;
  PicData& operator=(const PicData&) = delete;
  ~PicData() { release_staging(); }
  void release_staging() {
    if (stg) { hmgpu_staging_free(stg_ctx, stg); stg = nullptr; stg_ctx = nullptr; }
  }
  // the device context goes away while the application still holds the picture: the views become copies
  void detach_from_device() {
    if (!stg) return;
    for (auto* v : {&depth, &tr_idx, &cbf[0], &cbf[1], &cbf[2], &ts[0], &ts[1], &ts[2], &intra_dir[0], &intra_dir[1], &bypass, &ipcm}) v->detach();
    for (auto* v : {&part_size, &pred_mode, &qp, &ref_idx[0], &ref_idx[1]}) v->detach();
    mv[0].detach(); mv[1].detach(); slice_idx.detach(); tile_idx.detach();
    for (int c = 0; c < 3; c++) { coeff[c].detach(); level_start[c].detach(); }
    release_staging();
  }
  // gpu: the device context the picture will be handed to (its arrays live in a staging block of that context), or null
  void allocate(const Sps& sps, const ZScan* z, hmgpu_ctx* gpu = nullptr) {
    width = sps.width; height = sps.height; log2_ctb = sps.log2_ctb;
    ctbs_w = sps.pic_w_ctbs(); ctbs_h = sps.pic_h_ctbs(); num_ctbs = ctbs_w * ctbs_h;
    parts = 1 << (2 * log2_ctb - 4);
    zs = z;
    csx = sps.csx(); csy = sps.csy(); cshift = csx + csy;
    const size_t n = (size_t)num_ctbs * parts;
    const size_t luma = (size_t)num_ctbs << (2 * log2_ctb);
    const size_t chroma = luma >> cshift;
    if (gpu && hmgpu_staging_alloc(gpu, &stg, &stg_meta, &stg_co) == HMGPU_OK) {
      stg_ctx = gpu;
      const hmgpu_ctu_meta& m = stg_meta;
      depth.bind(m.depth, n); tr_idx.bind(m.tr_idx, n); part_size.bind(m.part_size, n); pred_mode.bind(m.pred_mode, n); qp.bind(m.qp, n);
      for (int c = 0; c < 3; c++) { cbf[c].bind(m.cbf[c], n); ts[c].bind(m.transform_skip[c], n); }
      for (int l = 0; l < 2; l++) { mv[l].bind(m.mv[l], 2 * n); ref_idx[l].bind(m.ref_idx[l], n); intra_dir[l].bind(m.intra_dir[l], n); }
      bypass.bind(m.transquant_bypass, n); ipcm.bind(m.ipcm, n);
      slice_idx.bind(m.slice_idx, num_ctbs); tile_idx.bind(m.tile_idx, num_ctbs);
      for (int c = 0; c < 3; c++) { coeff[c].bind(stg_co.level[c], c ? chroma : luma); level_start[c].bind(stg_co.ctu_level_start[c], num_ctbs + 1); }
      for (int c = 0; c < 3; c++) tu_off[c].assign(n, 0xffffffffu);
    } else {
      stg = nullptr;
      for (auto* v : {&depth, &tr_idx, &cbf[0], &cbf[1], &cbf[2], &ts[0], &ts[1], &ts[2], &intra_dir[0], &intra_dir[1], &bypass, &ipcm}) v->assign(n, 0);
      for (auto* v : {&part_size, &pred_mode, &qp, &ref_idx[0], &ref_idx[1]}) v->assign(n, 0);
      mv[0].assign(2 * n, 0);
      mv[1].assign(2 * n, 0);
      slice_idx.assign(num_ctbs, 0);
      tile_idx.assign(num_ctbs, 0);
      coeff[0].assign(luma, 0);
      coeff[1].assign(chroma, 0);
      coeff[2].assign(chroma, 0);
      for (int c = 0; c < 3; c++) level_start[c].assign(num_ctbs + 1, 0);
    }
    for (auto* v : {&skip, &merge, &merge_idx, &inter_dir}) v->assign(n, 0);
    slice_addr.assign(num_ctbs, -1);
    if (sps.pcm) { pcm[0].assign(luma, 0); pcm[1].assign(chroma, 0); pcm[2].assign(chroma, 0); }
    if (sps.chroma_format_idc == 3) { ccp[0].assign(n, 0); ccp[1].assign(n, 0); }
    sao.assign((size_t)num_ctbs * 3, hmgpu_sao_param{});
  }
  // a new picture in the same buffers.  Only the per-CTB bookkeeping is cleared here; the arrays of a CTU are brought to the
  // state HM's TComDataCU::initCtu leaves (TComDataCU.cpp:420-470) by reset_ctu() when the parser reaches the CTU -- on the
  // parser's thread, right before it writes them
  void reset() {
    std::fill(slice_idx.begin(), slice_idx.end(), 0);
    std::fill(slice_addr.begin(), slice_addr.end(), -1);
    slices.clear();
    slices.reserve(HMGPU_MAX_SLICES);   // entries are added while a parser thread reads earlier ones: the storage must not move
    has_pcm = has_bypass = decoded = filtered = planes_valid = false;
    level_cursor[0] = level_cursor[1] = level_cursor[2] = 0;
    hash_mismatch = false;
    dl_ticket = 0;
    sei_hash_method = 0;
  }
  void reset_ctu(int rs) {
    const size_t first = (size_t)rs * parts;
    auto fill = [&](auto& v, auto value) { std::fill(v.begin() + first, v.begin() + first + parts, value); };
    fill(depth, (uint8_t)0);
    fill(tr_idx, (uint8_t)0);
    for (int c = 0; c < 3; c++) { fill(cbf[c], (uint8_t)0); fill(ts[c], (uint8_t)0); }
    fill(intra_dir[0], (uint8_t)1);
    fill(intra_dir[1], (uint8_t)0);
    fill(bypass, (uint8_t)0); fill(ipcm, (uint8_t)0); fill(skip, (uint8_t)0); fill(merge, (uint8_t)0); fill(merge_idx, (uint8_t)0); fill(inter_dir, (uint8_t)0);
    if (!ccp[0].empty()) { fill(ccp[0], (int8_t)0); fill(ccp[1], (int8_t)0); }
    fill(part_size, (int8_t)HMGPU_SIZE_NONE);
    fill(pred_mode, (int8_t)2);          // HM's NUMBER_OF_PREDICTION_MODES: nothing decoded here
    fill(qp, (int8_t)0);
    for (int l = 0; l < 2; l++) {
      fill(ref_idx[l], (int8_t)-1);
      std::fill(mv[l].begin() + 2 * first, mv[l].begin() + 2 * (first + parts), (int16_t)0);
    }
    const size_t luma = (size_t)1 << (2 * log2_ctb);
    if (compact) {
      // the CTU's coded TUs will follow here (a CTU that is reset again after a parse error keeps no TU: whatever it wrote is dead space)
      for (int c = 0; c < 3; c++) { level_start[c][rs] = level_cursor[c]; std::fill(tu_off[c].begin() + first, tu_off[c].begin() + first + parts, 0xffffffffu); }
    } else {
      std::fill(coeff[0].begin() + rs * luma, coeff[0].begin() + (rs + 1) * luma, (int16_t)0);
      for (int c = 1; c < 3; c++) std::fill(coeff[c].begin() + rs * (luma >> cshift), coeff[c].begin() + (rs + 1) * (luma >> cshift), (int16_t)0);
    }
    for (int c = 0; c < 3; c++) sao[(size_t)rs * 3 + c] = hmgpu_sao_param{};
  }
  // where the levels of the TU of component c that starts at partition z of CTB ctb go (size x size, zeroed): HM's place, or the next free one
  // sub: the lower square of a 4:2:2 chroma block (dense layout only: it follows the upper one, TComTU::VERTICAL_SPLIT)
  int16_t* level_dst(int c, size_t ctb, size_t z, int size, int sub = 0) {
    if (!compact) return c == 0 ? &coeff[0][(ctb << (2 * log2_ctb)) + 16 * z] : &coeff[c][(ctb << (2 * log2_ctb - cshift)) + ((16 * z) >> cshift) + (size_t)sub * size * size];
    const uint32_t off = level_cursor[c];
    // (a damaged stream can deliver the same CTUs twice: what they wrote first is dead space, and the space is finite)
    if ((size_t)off + (size_t)size * size > coeff[c].size()) throw ParseError("more coded transform blocks than the picture has room for");
    level_cursor[c] += (uint32_t)(size * size);
    tu_off[c][ctb * parts + z] = off;
    int16_t* d = &coeff[c][off];
    memset(d, 0, (size_t)size * size * sizeof(int16_t));
    return d;
  }
  // compact -> HM's dense layout, into dense_levels[] (pictures of damaged streams whose CTUs did not arrive once and in order).  A TU's
  // size follows from its partition's depth and transform depth; only TUs of decoded CTUs that the arrays still call coded are taken
  HostVec<int16_t> dense_levels[3];
  void expand_dense() {
    const size_t luma = (size_t)num_ctbs << (2 * log2_ctb);
    dense_levels[0].assign(luma, 0); dense_levels[1].assign(luma >> cshift, 0); dense_levels[2].assign(luma >> cshift, 0);
    for (int c = 0; c < 3; c++)
      for (size_t ctb = 0; ctb < (size_t)num_ctbs; ctb++) {
        if (slice_addr[ctb] < 0) continue;
        for (size_t z = 0; z < (size_t)parts; z++) {
          const size_t part = ctb * parts + z;
          const uint32_t off = tu_off[c][part];
          if (off == 0xffffffffu) continue;
          const int log2tu = log2_ctb - depth[part] - tr_idx[part];
          int size = 1 << (c ? (log2tu > 2 ? log2tu - 1 : 2) : log2tu);
          if (size > 32) size = 32;
          const size_t dst = c == 0 ? (ctb << (2 * log2_ctb)) + 16 * z : (ctb << (2 * log2_ctb - 2)) + 4 * z;
          const size_t n = (size_t)size * size;
          if (!((cbf[c][part] >> tr_idx[part]) & 1) || off + n > coeff[c].size() || dst + n > dense_levels[c].size()) continue;
          memcpy(&dense_levels[c][dst], &coeff[c][off], n * sizeof(int16_t));
        }
      }
  }
  // the levels of the TU that starts at partition z of CTB ctb, or null (compact: no coded TU starts there)
  const int16_t* level_src(int c, size_t ctb, size_t z) const {
    if (!compact) return c == 0 ? &coeff[0][(ctb << (2 * log2_ctb)) + 16 * z] : &coeff[c][(ctb << (2 * log2_ctb - cshift)) + ((16 * z) >> cshift)];
    const uint32_t off = tu_off[c][ctb * parts + z];
    return off == 0xffffffffu ? nullptr : &coeff[c][off];
  }
  size_t part_at(int x, int y) const {      // partition index of the 4x4 block covering luma sample (x, y)
    const int mask = (1 << log2_ctb) - 1;
    return (size_t)((y >> log2_ctb) * ctbs_w + (x >> log2_ctb)) * parts + zs->r2z[((y & mask) >> 2) * zs->n4 + ((x & mask) >> 2)];
  }
  int ctb_at(int x, int y) const { return (y >> log2_ctb) * ctbs_w + (x >> log2_ctb); }
};

}  // namespace hmdec
