// decoder.cpp -- see decoder.h.  Clause numbers refer to Rec. ITU-T H.265.
#include "decoder.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <pthread.h>
#include <new>

#include "md5.h"

namespace hmdec {

// ---- host memory for the picture arrays: page-locked through libhmgpu when a device is in use, malloc otherwise.  A 64-byte
// header in front of every block remembers which of the two it came from.
namespace {
thread_local bool tl_pinned = false;
}
void host_alloc_use_pinned(bool on) { tl_pinned = on; }
void* host_alloc(size_t bytes) {
  uint8_t* base = nullptr;
  bool pinned = false;
  if (tl_pinned) { base = static_cast<uint8_t*>(hmgpu_host_alloc(bytes + 64)); pinned = base != nullptr; }
  if (!base) base = static_cast<uint8_t*>(malloc(bytes + 64));
  if (!base) throw std::bad_alloc();
  base[0] = pinned ? 1 : 0;
  return base + 64;
}
void host_free(void* p) {
  if (!p) return;
  uint8_t* base = static_cast<uint8_t*>(p) - 64;
  if (base[0]) hmgpu_host_free(base); else free(base);
}

static bool is_irap(int t) { return t >= NAL_BLA_W_LP && t <= NAL_RSV_IRAP_VCL23; }
static bool is_idr(int t) { return t == NAL_IDR_W_RADL || t == NAL_IDR_N_LP; }
static bool is_bla(int t) { return t >= NAL_BLA_W_LP && t <= NAL_BLA_N_LP; }
static bool is_rasl(int t) { return t == NAL_RASL_N || t == NAL_RASL_R; }
static bool is_radl(int t) { return t == NAL_RADL_N || t == NAL_RADL_R; }
static bool is_sub_layer_non_ref(int t) { return t < 16 && (t & 1) == 0; }

Decoder::Decoder() { memset(pending_hash_val_, 0, sizeof(pending_hash_val_)); }

Decoder::~Decoder() {
  if (!hash_threads_.empty()) {
    drain_hash_jobs();
    { std::lock_guard<std::mutex> lk(hash_mu_); hash_stop_ = true; }
    hash_cv_.notify_all();
    for (std::thread& t : hash_threads_) t.join();
  }
  if (!workers_.empty()) {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_work_.notify_all();
    cv_progress_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  if (gpu_ && getenv("HMDEC_STATS")) {          // tuning aid: device time per kernel class over the life of the decoder
    for (size_t d = 0; d < gpus_.size(); d++) {
      hmgpu_stats st;
      if (hmgpu_get_stats(gpus_[d], &st, 0) == HMGPU_OK)
        for (int k = 0; k < HMGPU_NUM_KERNELS; k++)
          if (st.kernel_launches[k]) fprintf(stderr, "hmdec[%zu]: %-14s %8.3f ms in %llu launches\n", d, hmgpu_kernel_name(k), st.kernel_ms[k], (unsigned long long)st.kernel_launches[k]);
    }
    if (gpus_.size() > 1) fprintf(stderr, "hmdec: %llu bytes of reference pictures copied between the contexts\n", (unsigned long long)transfer_bytes());
  }
  poll_device_hashes(true);
  batch_.clear();
  pool_.clear();                            // (the pictures' staging blocks belong to the first context)
  retired_.clear();
  for (size_t d = gpus_.size(); d-- > 0;) hmgpu_destroy(gpus_[d]);
}

uint64_t Decoder::transfer_bytes() const {
  uint64_t n = transfer_bytes_closed_;
  for (hmgpu_ctx* g : gpus_) n += hmgpu_transfer_bytes(g);
  return n;
}

void Decoder::sync_all() {
  for (hmgpu_ctx* g : gpus_) {
    const hmgpu_status st = hmgpu_sync(g);
    if (st != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_sync: ") + hmgpu_status_string(st));   // e.g. an intra wavefront that gave up
  }
  synced_seq_ = submitted_seq_;
}

// ------------------------------------------------------------------------------------------------ NAL level
bool Decoder::push(const uint8_t* d, size_t len, int max_tl, int* nal_type_out) {
  const bool again = push_unit(d, len, max_tl, nal_type_out);
  // a picture further back in the parser pipeline failed: said now that this unit has been processed (when the unit has to come
  // again -- "new picture" -- the report waits for the repetition)
  if (!again && !deferred_error_.empty()) throw ParseError(take_deferred_error());
  return again;
}

bool Decoder::push_unit(const uint8_t* d, size_t len, int max_tl, int* nal_type_out) {
  if (len >= 3 && d[0] == 0 && d[1] == 0 && d[2] == 1) { d += 3; len -= 3; }
  else if (len >= 4 && d[0] == 0 && d[1] == 0 && d[2] == 0 && d[3] == 1) { d += 4; len -= 4; }
  if (len < 2) throw ParseError("NAL unit shorter than its header");
  if (d[0] & 0x80) throw ParseError("forbidden_zero_bit is set");
  const int type = (d[0] >> 1) & 0x3f, layer = ((d[0] & 1) << 5) | (d[1] >> 3), tid = (d[1] & 7) - 1;
  if (nal_type_out) *nal_type_out = type;
  if (tid < 0) throw ParseError("nuh_temporal_id_plus1 is zero");
  if (layer != 0 || (max_tl >= 0 && tid > max_tl)) return false;
  for (auto& p : pool_) p->lent = false;
  max_tl_ = max_tl;
  if (threaded()) retire_ready(false);
  if (repush_pending_) {                   // the unit that was answered with "new picture" comes again, as the protocol demands
    repush_pending_ = false;
    if (len == repush_len_ && memcmp(d, repush_head_, std::min<size_t>(len, sizeof(repush_head_))) == 0) return false;
  }
  bool taken_early = false;
  if (type <= NAL_RASL_R || (type >= NAL_BLA_W_LP && type <= NAL_CRA)) {
    if (len < 3) throw ParseError("slice NAL unit without payload");
    if ((d[2] & 0x80) && cur_) {           // first_slice_segment_in_pic_flag while a picture is open: close it, unit comes again
      if (!threaded()) { finish_picture(); return true; }
      // A unit that activates another sequence parameter set replaces the picture store (activate()): it must not be taken in before
      // the application has fetched what is still to be output.  Answer "new picture" with the pipeline drained, exactly as the
      // single-threaded decoder does; the unit is decoded when it comes again.
      bool reseq = true;
      try {
        std::vector<uint8_t> probe = nal_to_rbsp(d + 2, len - 2, nullptr);
        BitReader pbr(probe.data(), probe.size());
        SliceHeader psh;
        parse_slice_header(pbr, type, tid, ps_, have_independent_ ? &last_independent_ : nullptr, psh);
        reseq = opens_new_sequence(psh);
      } catch (...) {}                     // (an unparsable header is reported when the unit comes again)
      if (reseq) { finish_picture(); return true; }
      // Parser threads: the finished picture is closed (its thread may still be busy: nobody waits here) and the new unit is taken
      // in at once, so that its parsing overlaps whatever the caller does before repeating it -- typically fetching output
      // pictures, which waits for the device.  The answer is still "new picture"; the repetition is skipped above.
      close_current();
      retire_ready(false);
      taken_early = true;
      repush_len_ = len;
      memset(repush_head_, 0, sizeof(repush_head_));
      memcpy(repush_head_, d, std::min<size_t>(len, sizeof(repush_head_)));
    }
    std::vector<size_t> epb;
    std::vector<uint8_t> rbsp = nal_to_rbsp(d + 2, len - 2, &epb);
    BitReader br(rbsp.data(), rbsp.size());
    SliceHeader sh;
    parse_slice_header(br, type, tid, ps_, have_independent_ ? &last_independent_ : nullptr, sh);
    decode_slice(rbsp, br, sh, epb);
    if (taken_early) { repush_pending_ = true; return true; }
    return false;
  }
  std::vector<uint8_t> rbsp = nal_to_rbsp(d + 2, len - 2);
  BitReader br(rbsp.data(), rbsp.size());
  switch (type) {
    case NAL_VPS: { auto v = parse_vps(br); ps_.vps[v->id] = v; break; }
    case NAL_SPS: { auto s = parse_sps(br); ps_.sps[s->id] = s; break; }
    case NAL_PPS: { auto p = parse_pps(br); ps_.pps[p->id] = p; break; }
    case NAL_PREFIX_SEI: parse_sei(rbsp, false); break;
    case NAL_SUFFIX_SEI: parse_sei(rbsp, true); break;
    case NAL_EOS:
      finish_picture();
      after_eos_ = true;
      break;
    default: break;                        // AUD, EOB, filler data, reserved and unspecified types carry nothing to decode
  }
  return false;
}

// D.2.1 sei_message(); only the decoded picture hash (D.2.19, payloadType 132) is of interest
void Decoder::parse_sei(const std::vector<uint8_t>& rbsp, bool suffix) {
  BitReader br(rbsp.data(), rbsp.size());
  while (br.more_rbsp_data()) {
    int type = 0, size = 0;
    for (;;) { const int b = br.u(8); type += b; if (b != 255) break; }
    for (;;) { const int b = br.u(8); size += b; if (b != 255) break; }
    const size_t end = br.pos() + 8 * (size_t)size;
    if (end > br.size_bits()) throw ParseError("SEI payload runs past the NAL unit");
    if (suffix && type == 132 && size >= 1) {
      const int hash_type = br.u(8);
      const int bytes = hash_type == 0 ? 16 : hash_type == 1 ? 2 : hash_type == 2 ? 4 : 0;
      const int ncomp = (sps_ && sps_->chroma_format_idc == 0) ? 1 : 3;      // one digest per colour component of the format (D.2.19)
      if (bytes && size >= 1 + ncomp * bytes) {
        memset(pending_hash_val_, 0, sizeof(pending_hash_val_));
        for (int c = 0; c < ncomp; c++) for (int i = 0; i < bytes; i++) pending_hash_val_[c][i] = (uint8_t)br.u(8);
        pending_hash_method_ = hash_type + 1;
        pending_hash_ = true;
        if (cur_) {                        // the SEI follows the slices of the picture it describes
          memcpy(cur_->sei_hash, pending_hash_val_, sizeof(pending_hash_val_));
          cur_->sei_hash_method = pending_hash_method_;
          pending_hash_ = false;
        }
      }
    }
    br.skip(end - br.pos());
  }
}

// ------------------------------------------------------------------------------------------------ picture level
// the slice's parameter sets need another picture store / device context than the active ones
bool Decoder::opens_new_sequence(const SliceHeader& sh) const {
  if (sh.pps_id < 0 || sh.pps_id >= 64 || !ps_.pps[sh.pps_id]) return true;
  const std::shared_ptr<Pps>& pps = ps_.pps[sh.pps_id];
  if (pps->sps_id < 0 || pps->sps_id >= 16 || !ps_.sps[pps->sps_id]) return true;
  const std::shared_ptr<Sps>& sps = ps_.sps[pps->sps_id];
  return !sps_ || sps_->width != sps->width || sps_->height != sps->height || sps_->log2_ctb != sps->log2_ctb ||
         sps_->bit_depth_luma != sps->bit_depth_luma || sps_->bit_depth_chroma != sps->bit_depth_chroma || sps_->pcm != sps->pcm ||
         sps_->pcm_bit_depth_luma != sps->pcm_bit_depth_luma || sps_->pcm_bit_depth_chroma != sps->pcm_bit_depth_chroma ||
         sps_->pcm_loop_filter_disabled != sps->pcm_loop_filter_disabled || sps_->strong_intra_smoothing != sps->strong_intra_smoothing ||
         sps_->range_ext_flags() != sps->range_ext_flags() ||
         sps_->max_dec_pic_buffering[sps_->max_sub_layers - 1] != sps->max_dec_pic_buffering[sps->max_sub_layers - 1];
}

void Decoder::activate(const SliceHeader& sh) {
  std::shared_ptr<Pps> pps = ps_.pps[sh.pps_id];
  std::shared_ptr<Sps> sps = ps_.sps[pps->sps_id];
  const bool new_seq = opens_new_sequence(sh);
  if (new_seq && threaded()) { close_current(); retire_ready(true); }      // nothing of the old sequence may still be in flight
  sps_ = sps;
  pps_ = pps;
  // (a PPS object is shared with the parser threads of earlier pictures: its tables are derived once per picture geometry)
  if (pps_->derived_w != sps_->pic_w_ctbs() || pps_->derived_h != sps_->pic_h_ctbs()) {
    // a PPS object whose tables were derived for another geometry may be in use by pictures still being parsed: wait for them.  A PPS
    // that has just arrived (parameter sets are commonly repeated in front of every IRAP picture) is not shared with anybody yet.
    if (threaded() && pps_->derived_w >= 0) { close_current(); retire_ready(true); }
    pps_->derive_tiles(*sps_);
    pps_->derived_w = sps_->pic_w_ctbs();
    pps_->derived_h = sps_->pic_h_ctbs();
  }
  if (!new_seq) return;
  if (!is_irap(sh.nal_type)) throw ParseError("a new sequence parameter set is activated by a picture that is not an IRAP picture");
  // a new coded video sequence with another geometry: the picture store starts over.  Pictures the application holds or has still
  // to fetch from the output queue stay alive (retired_, samples on the host) until the next change of sequence; pictures that were
  // never put out are dropped, as in HM.
  flush_batch();
  drain_hash_jobs();                       // (the hash threads read the planes of the pictures that go away here)
  retired_.clear();
  for (auto& p : pool_) {
    const bool queued = std::find(out_queue_.begin(), out_queue_.end(), p.get()) != out_queue_.end();
    if (queued) fetch_planes(p.get());     // (the device context that holds the samples goes away below)
    if (queued || p->lent) { p->detach_from_device(); retired_.push_back(std::move(p)); }
  }
  pool_.clear();
  scan_.clear();
  scan_idx_ = 0;
  last_decoded_ = nullptr;
  transfer_bytes_closed_ = transfer_bytes();
  for (size_t d = gpus_.size(); d-- > 0;) hmgpu_destroy(gpus_[d]);
  gpus_.clear();
  gpu_ = nullptr;
  zscan_.init(sps_->log2_ctb);
  memset(&seq_, 0, sizeof(seq_));
  seq_.width = sps_->width;
  seq_.height = sps_->height;
  seq_.bit_depth_luma = sps_->bit_depth_luma;
  seq_.bit_depth_chroma = sps_->bit_depth_chroma;
  seq_.chroma_format = sps_->chroma_format_idc;            // 1, or 0: monochrome (chroma planes exist on the device and are never shown)
  seq_.log2_ctu_size = sps_->log2_ctb;
  seq_.max_pictures = std::min(40, sps_->max_dec_pic_buffering[sps_->max_sub_layers - 1] + 3 + (threaded() ? threads_ + 1 : 0));
  seq_.pcm_loop_filter_disable = sps_->pcm && sps_->pcm_loop_filter_disabled;
  seq_.strong_intra_smoothing = sps_->strong_intra_smoothing;
  seq_.range_ext_flags = sps_->range_ext_flags();
  seq_.pcm_bit_depth_luma = sps_->pcm_bit_depth_luma;
  seq_.pcm_bit_depth_chroma = sps_->pcm_bit_depth_chroma;
  if (!parse_only_) {
    for (int ordinal : devices_) {
      hmgpu_ctx* g = nullptr;
      const hmgpu_status st = hmgpu_create(&seq_, ordinal, &g);
      if (st != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_create: ") + hmgpu_status_string(st));
      gpus_.push_back(g);
      if (getenv("HMDEC_STATS")) hmgpu_set_profiling(g, 1);
    }
    gpu_ = gpus_[0];
  }
}

PicData* Decoder::acquire_buffer() {
  // a buffer nobody needs any more; `pinned_too`: also one whose only holders are readers that let go by themselves (parser tasks
  // that predict from it, hash jobs that read its planes)
  auto reusable = [&](const PicData* p, bool pinned_too) {
    return p != cur_ && !p->is_reference && !p->needed_for_output && !p->lent && !p->in_flight && (pinned_too || p->users.load() == 0);
  };
  for (int attempt = 0; threaded() && attempt < 4096; attempt++) {
    bool free_one = (int)pool_.size() < seq_.max_pictures, pinned_one = false;
    for (auto& p : pool_) { free_one |= reusable(p.get(), false); pinned_one |= reusable(p.get(), true); }
    if (free_one) break;
    if (!inflight_.empty()) {
      // every buffer is held by a picture still in flight: wait for the oldest one, hand it on, look again
      { std::unique_lock<std::mutex> lk(mu_); PicData* head = inflight_.front()->pic; cv_progress_.wait(lk, [&] { return head->parse_done.load(); }); }
      retire_ready(false);
      continue;
    }
    if (!pinned_one || hash_threads_.empty()) break;     // nothing will come free by itself: the stream holds more pictures than the SPS allows
    // the pipeline is empty and the only reusable buffers are still being hashed (MD5 behind decoding: few cores, large pictures):
    // throttle instead of failing.  The hash threads release a picture (users) before they take hash_mu_ and notify.
    std::unique_lock<std::mutex> lk(hash_mu_);
    hash_idle_cv_.wait(lk, [&] { for (auto& p : pool_) if (reusable(p.get(), false)) return true; return hash_jobs_.empty() && hash_busy_ == 0; });
  }
  for (auto& p : pool_)
    if (reusable(p.get(), false)) {
      flush_batch();                                      // (nothing that names the buffer may still be waiting to be submitted)
      if (gpu_ && p->submit_seq > synced_seq_) {          // the copies out of the arrays of the picture that lived here may still be under way
        if (p->stg && !p->has_pcm) {                         // (PCM samples travel from arrays of their own)
          const hmgpu_status st = hmgpu_staging_wait(gpu_, p->stg);   // the copies only (of whichever context read the block): the kernels read the device's arrays
          if (st != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_staging_wait: ") + hmgpu_status_string(st));
        } else {
          sync_all();
        }
      }
      return p.get();
    }
  if ((int)pool_.size() >= seq_.max_pictures) throw ParseError("decoded picture buffer overflow (more pictures held than the SPS allows)");
  pool_.emplace_back(new PicData());
  PicData* p = pool_.back().get();
  host_alloc_use_pinned(gpu_ != nullptr);
  p->allocate(*sps_, &zscan_, gpu_);
  if (gpu_) for (int c = 0; c < 3; c++) p->plane[c].resize((size_t)(p->width >> (c ? p->csx : 0)) * (p->height >> (c ? p->csy : 0)));
  host_alloc_use_pinned(false);
  for (size_t d = 0; d < gpus_.size(); d++) {
    // the same buffer in every context: acquired in the same order everywhere, so the handles -- what ref_pic[][] of the slice parameters
    // names -- are the same numbers
    hmgpu_pic h = HMGPU_NO_PIC;
    const hmgpu_status st = hmgpu_picture_acquire(gpus_[d], &h);
    if (st != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_picture_acquire: ") + hmgpu_status_string(st));
    if (d == 0) p->handle = h;
    else if (h != p->handle) throw std::runtime_error("hmdec: the device contexts do not hand out the same picture handles");
    if (d > 0 && p->stg) {
      const hmgpu_status ss = hmgpu_staging_share(gpu_, p->stg, gpus_[d]);
      if (ss != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_staging_share: ") + hmgpu_status_string(ss));
    }
  }
  return p;
}

// 8.3.1
int Decoder::compute_poc(const SliceHeader& sh) {
  if (is_idr(sh.nal_type)) return 0;
  const int max_lsb = 1 << sps_->log2_max_poc_lsb;
  int msb = 0;
  if (!(is_irap(sh.nal_type) && no_rasl_output_)) {
    const int prev_lsb = prev_tid0_poc_ & (max_lsb - 1), prev_msb = prev_tid0_poc_ - prev_lsb;
    if (sh.poc_lsb < prev_lsb && prev_lsb - sh.poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
    else if (sh.poc_lsb > prev_lsb && sh.poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
    else msb = prev_msb;
  }
  return msb + sh.poc_lsb;
}

PicData* Decoder::find_ref(int poc, bool lsb_only, bool any_marking) {
  const int mask = (1 << sps_->log2_max_poc_lsb) - 1;
  for (auto& p : pool_) {
    if (p.get() == cur_ || !(p->decoded || p->in_flight)) continue;
    if (!p->is_reference && !any_marking) continue;
    if (lsb_only ? (p->poc & mask) == poc : p->poc == poc) return p.get();
  }
  return nullptr;
}

// 8.3.2 decoding process for the reference picture set
void Decoder::apply_rps(const SliceHeader& sh) {
  st_before_.clear();
  st_after_.clear();
  lt_curr_.clear();
  if (is_irap(sh.nal_type) && no_rasl_output_)
    for (auto& p : pool_) if (p.get() != cur_) p->is_reference = p->is_long_term = false;
  if (is_idr(sh.nal_type)) return;
  const int max_lsb = 1 << sps_->log2_max_poc_lsb;
  std::vector<PicData*> keep;
  // long-term pictures first: a picture named as long-term must not be picked up as short-term (8.3.2)
  std::vector<PicData*> lt_all;
  for (int i = 0; i < sh.num_long_term; i++) {
    int poc = sh.lt_poc[i];
    if (sh.lt_msb_present[i]) poc += cur_->poc - sh.lt_msb_cycle[i] * max_lsb - (cur_->poc & (max_lsb - 1));
    PicData* p = find_ref(poc, !sh.lt_msb_present[i], false);
    if (p) { keep.push_back(p); lt_all.push_back(p); }
    if (sh.lt_used[i]) {
      if (!p) throw ParseError("long-term reference picture is missing from the decoded picture buffer");
      lt_curr_.push_back(p);
    }
  }
  for (int i = 0; i < sh.rps.num_delta_pocs(); i++) {
    const int poc = cur_->poc + sh.rps.delta_poc[i];
    PicData* p = nullptr;
    for (auto& q : pool_)
      if (q.get() != cur_ && (q->decoded || q->in_flight) && q->is_reference && !q->is_long_term && q->poc == poc && std::find(lt_all.begin(), lt_all.end(), q.get()) == lt_all.end()) p = q.get();
    if (p) keep.push_back(p);
    if (sh.rps.used[i]) {
      if (!p) throw ParseError("short-term reference picture is missing from the decoded picture buffer");
      (i < sh.rps.num_negative ? st_before_ : st_after_).push_back(p);
    }
  }
  for (PicData* p : lt_all) p->is_long_term = true;
  for (auto& p : pool_)
    if (p.get() != cur_ && p->is_reference && std::find(keep.begin(), keep.end(), p.get()) == keep.end()) p->is_reference = p->is_long_term = false;
}

// 8.3.4 decoding process for reference picture lists construction
void Decoder::build_ref_lists(const SliceHeader& sh, SliceInfo& si) {
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < 16; i++) { si.ref_pics[l][i] = nullptr; si.ref_poc[l][i] = 0; si.ref_is_lt[l][i] = false; }
  if (sh.type == SLICE_I) return;
  const int total = (int)(st_before_.size() + st_after_.size() + lt_curr_.size());
  if (total == 0) throw ParseError("P or B slice with an empty reference picture set");
  for (int l = 0; l < (sh.type == SLICE_B ? 2 : 1); l++) {
    const std::vector<PicData*>& a = l == 0 ? st_before_ : st_after_;
    const std::vector<PicData*>& b = l == 0 ? st_after_ : st_before_;
    std::vector<PicData*> temp;
    std::vector<bool> lt;
    const int n = std::max(sh.num_ref_idx[l], total);
    while ((int)temp.size() < n) {
      for (size_t i = 0; i < a.size() && (int)temp.size() < n; i++) { temp.push_back(a[i]); lt.push_back(false); }
      for (size_t i = 0; i < b.size() && (int)temp.size() < n; i++) { temp.push_back(b[i]); lt.push_back(false); }
      for (size_t i = 0; i < lt_curr_.size() && (int)temp.size() < n; i++) { temp.push_back(lt_curr_[i]); lt.push_back(true); }
    }
    for (int i = 0; i < sh.num_ref_idx[l]; i++) {
      const int k = sh.list_mod_flag[l] ? sh.list_entry[l][i] : i;
      si.ref_pics[l][i] = temp[k];
      si.ref_poc[l][i] = temp[k]->poc;
      si.ref_is_lt[l][i] = lt[k];
    }
  }
}

// the per-slice constants of include/hmgpu.h (TComSlice getters the hot path reads)
void Decoder::build_slice_params(const SliceHeader& sh, SliceInfo& si) {
  hmgpu_slice_params& p = si.params;
  memset(&p, 0, sizeof(p));
  si.type = sh.type;
  si.header = sh;
  p.slice_type = sh.type;
  p.cb_qp_offset = pps_->cb_qp_offset + sh.cb_qp_offset;
  p.cr_qp_offset = pps_->cr_qp_offset + sh.cr_qp_offset;
  p.pps_cb_qp_offset = pps_->cb_qp_offset;
  p.pps_cr_qp_offset = pps_->cr_qp_offset;
  p.deblocking_disable = sh.deblocking_disabled;
  p.beta_offset_div2 = sh.beta_offset_div2;
  p.tc_offset_div2 = sh.tc_offset_div2;
  p.lf_across_slices = sh.lf_across_slices;
  p.weighted_pred = (sh.type == SLICE_P && pps_->weighted_pred) || (sh.type == SLICE_B && pps_->weighted_bipred);
  p.lf_across_tiles = pps_->lf_across_tiles;
  p.constrained_intra_pred = pps_->constrained_intra_pred;
  for (int l = 0; l < 2; l++) {
    p.num_ref_idx[l] = sh.num_ref_idx[l];
    for (int i = 0; i < HMGPU_MAX_REF; i++) {
      p.ref_pic[l][i] = i < sh.num_ref_idx[l] && si.ref_pics[l][i] ? si.ref_pics[l][i]->handle : HMGPU_NO_PIC;
      p.ref_poc[l][i] = si.ref_poc[l][i];
    }
  }
  if (p.weighted_pred) {                     // TComSlice::initWpScaling (TComSlice.cpp:1476-1510): offsets at the coding bit depth
    p.wp_log2_denom[0] = sh.luma_log2_weight_denom;
    p.wp_log2_denom[1] = sh.chroma_log2_weight_denom;
    for (int l = 0; l < 2; l++)
      for (int i = 0; i < sh.num_ref_idx[l]; i++) {
        const PredWeight& w = sh.pw[l][i];
        p.wp_weight[l][i][0] = (int16_t)w.luma_weight;
        const bool hp = sps_->rext_high_precision_offsets;      // offsets already at the bit depth (TComWeightPrediction.cpp:257)
        p.wp_offset[l][i][0] = (int16_t)(w.luma_offset * (hp ? 1 : 1 << (sps_->bit_depth_luma - 8)));
        for (int c = 0; c < 2; c++) {
          p.wp_weight[l][i][1 + c] = (int16_t)w.chroma_weight[c];
          p.wp_offset[l][i][1 + c] = (int16_t)(w.chroma_offset[c] * (hp ? 1 : 1 << (sps_->bit_depth_chroma - 8)));
        }
      }
  }
  if (sps_->scaling_list_enabled) {           // TDecTop.cpp:651-668: PPS lists, else SPS lists (explicit or default)
    const ScalingListSet& src = pps_->scaling_list_data_present ? pps_->scaling_lists : sps_->scaling_lists;
    si.scaling_lists.reset(new hmgpu_scaling_lists);
    memcpy(si.scaling_lists->coef, src.coef, sizeof(src.coef));
    memcpy(si.scaling_lists->dc, src.dc, sizeof(src.dc));
    p.scaling_lists = si.scaling_lists.get();
  }
}

void Decoder::start_picture(const SliceHeader& sh) {
  cur_ = acquire_buffer();
  cur_->reset();
  // levels in the compact form when one parser walks the picture from front to back (wavefront rows / tiles may be shared out among threads)
  // (4:2:2 / 4:4:4 pictures keep HM's dense layout: the lower square of a 4:2:2 chroma block has its place behind the upper one)
  cur_->compact = gpu_ && cur_->stg && !pps_->entropy_coding_sync && !pps_->tiles_enabled && sps_->chroma_format_idc <= 1 && !getenv("HMDEC_DENSE_LEVELS");
  cur_->chroma_format = sps_->chroma_format_idc;
  cur_->poc = sh.poc;
  cur_->nal_type = sh.nal_type;
  cur_->temporal_id = sh.temporal_id;
  cur_->sao_enabled = sps_->sao;
  cur_->bit_depth[0] = sps_->bit_depth_luma; cur_->bit_depth[1] = sps_->bit_depth_chroma;
  cur_->pcm_bit_depth[0] = sps_->pcm_bit_depth_luma; cur_->pcm_bit_depth[1] = sps_->pcm_bit_depth_chroma;
  cur_->pcm_lf_disable = sps_->pcm && sps_->pcm_loop_filter_disabled;
  cur_->strong_intra = sps_->strong_intra_smoothing;
  cur_->range_ext_flags = sps_->range_ext_flags();
  cur_->num_comps = sps_->chroma_format_idc == 0 ? 1 : 3;
  cur_->sao_offset_shift[0] = pps_->sao_offset_shift[0]; cur_->sao_offset_shift[1] = pps_->sao_offset_shift[1];
  cur_->lf_across_tiles = pps_->lf_across_tiles;
  cur_->conf_window[0] = sps_->conf_left; cur_->conf_window[1] = sps_->conf_right; cur_->conf_window[2] = sps_->conf_top; cur_->conf_window[3] = sps_->conf_bottom;
  cur_->is_reference = true;                 // "used for short-term reference" until a later RPS says otherwise (8.3.1 end)
  cur_->is_long_term = false;
  cur_->pic_output = sh.pic_output && !(is_rasl(sh.nal_type) && skip_rasl_);
  cur_->needed_for_output = false;
  parse_state_ = PicParseState();
  if (pending_hash_) pending_hash_ = false;  // a hash SEI ahead of its picture does not occur (suffix SEI)
  apply_rps(sh);
  if (threaded()) {
    // at most threads_ pictures are parsed at once: wait for one of them before opening another
    for (;;) {
      int busy = 0;
      for (auto& t : inflight_) if (!t->pic->parse_done.load()) busy++;
      if (busy < threads_) break;
      std::unique_lock<std::mutex> lk(mu_);
      cv_progress_.wait(lk, [&] { for (auto& t : inflight_) if (t->pic->parse_done.load()) return true; return false; });
      lk.unlock();
      retire_ready(false);
    }
    cur_->rows_done.store(0);
    cur_->parse_done.store(false);
    cur_->in_flight = true;
    std::unique_ptr<PicTask> t(new PicTask());
    t->pic = cur_;
    cur_task_ = t.get();
    {
      std::lock_guard<std::mutex> lk(mu_);
      inflight_.push_back(std::move(t));
      runnable_.push_back(cur_task_);
    }
    cv_work_.notify_all();
  }
}

void Decoder::decode_slice(const std::vector<uint8_t>& rbsp, BitReader& br, SliceHeader& sh, const std::vector<size_t>& epb) {
  (void)br;
  if (sh.first_slice_segment_in_pic) {
    // 8.1: NoRaslOutputFlag of an IRAP picture: first picture, after an end of sequence, IDR or BLA
    if (is_irap(sh.nal_type)) {
      no_rasl_output_ = first_picture_ || after_eos_ || is_idr(sh.nal_type) || is_bla(sh.nal_type);
      if (no_rasl_output_) skip_rasl_ = !is_idr(sh.nal_type);
    } else if (first_picture_ || after_eos_) {
      return;                                // nothing decodable before the first IRAP picture (TDecTop::isRandomAccessSkipPicture)
    }
    activate(sh);
    sh.poc = compute_poc(sh);
    if (is_irap(sh.nal_type) && no_rasl_output_) poc_cra_ = sh.poc;
    if (is_rasl(sh.nal_type) && skip_rasl_ && sh.poc < poc_cra_) { have_independent_ = false; return; }   // associated IRAP starts the stream: skipped
    if (!is_rasl(sh.nal_type) && !is_radl(sh.nal_type) && !is_irap(sh.nal_type) && sh.poc > poc_cra_) skip_rasl_ = false;
    first_picture_ = false;
    after_eos_ = false;
    try {
      start_picture(sh);
    } catch (...) {                        // e.g. a reference picture is missing: the picture is not decoded at all
      if (cur_ && !cur_task_) { cur_->is_reference = cur_->is_long_term = cur_->needed_for_output = cur_->in_flight = false; cur_ = nullptr; }
      else if (cur_task_) close_current();
      throw;
    }
    if (sh.temporal_id == 0 && !is_rasl(sh.nal_type) && !is_radl(sh.nal_type) && !is_sub_layer_non_ref(sh.nal_type)) prev_tid0_poc_ = sh.poc;
  } else {
    if (!cur_) return;                       // slice of a picture that was skipped
    sh.poc = cur_->poc;
  }
  if (!sh.dependent) {
    last_independent_ = sh;
    have_independent_ = true;
    if (cur_->slices.size() >= HMGPU_MAX_SLICES) throw Unsupported("more slices in a picture than the device slice table holds");
    cur_->slices.emplace_back(new SliceInfo());
    SliceInfo& si = *cur_->slices.back();
    si.first_ctb_ts = pps_->ctb_rs_to_ts[sh.segment_address];
    try {
      build_ref_lists(sh, si);
      build_slice_params(sh, si);
    } catch (...) {
      cur_->slices.pop_back();             // the slice is not decoded; whatever the picture already holds stays
      have_independent_ = false;
      throw;
    }
  } else if (cur_->slices.empty()) {
    throw ParseError("dependent slice segment at the start of a picture");
  }
  if (!threaded()) {
    SliceDecoder sd(*sps_, *pps_, *cur_, parse_state_);
    sd.decode(sh, (int)cur_->slices.size() - 1, rbsp.data(), rbsp.size());
    return;
  }
  // the parser thread of this picture reads the motion data of its reference pictures: they stay until it is done
  const SliceInfo& si = *cur_->slices.back();
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < 16; i++) {
      PicData* r = si.ref_pics[l][i];
      if (r && std::find(cur_task_->held.begin(), cur_task_->held.end(), r) == cur_task_->held.end()) { r->users.fetch_add(1); cur_task_->held.push_back(r); }
    }
  SliceJob job;
  job.sh = sh;
  job.slice_idx = (int)cur_->slices.size() - 1;
  job.rbsp = rbsp;
  job.sps = sps_;
  job.pps = pps_;
  // a wavefront-coded picture in one slice segment with one entry point per CTB row: its rows can be parsed side by side
  const int rows = cur_->ctbs_h, W = cur_->ctbs_w;
  const int num_tiles = pps_->num_tile_cols * pps_->num_tile_rows;
  const bool by_rows = pps_->entropy_coding_sync && !pps_->tiles_enabled && rows > 1 && (int)sh.entry_points.size() == rows - 1;
  const bool by_tiles = pps_->tiles_enabled && !pps_->entropy_coding_sync && num_tiles > 1 && (int)sh.entry_points.size() == num_tiles - 1;
  if ((by_rows || by_tiles) && !sh.dependent && sh.segment_address == 0) {
    auto w = std::make_shared<WppShared>();
    const int units = by_rows ? rows : num_tiles;              // sub-streams: CTB rows, or tiles (which depend on nothing)
    w->init(units, W);
    w->tiles = by_tiles;
    // entry points count bytes of the NAL unit, emulation prevention bytes included; positions in the RBSP do not (7.4.7.1)
    const size_t rbsp_start = sh.data_bit_offset / 8;
    size_t nal_pos = rbsp_start;
    for (size_t i = 0; i < epb.size(); i++) if (epb[i] - i <= rbsp_start) nal_pos = rbsp_start + i + 1;
    bool ok = true;
    for (int r = 0; r < units; r++) {
      size_t removed = 0;
      while (removed < epb.size() && epb[removed] < nal_pos) removed++;
      const size_t pos = nal_pos - removed;
      if (pos >= rbsp.size()) { ok = false; break; }
      w->row_bit_pos[r] = pos * 8;
      if (r + 1 < units) nal_pos += sh.entry_points[r];
    }
    if (ok) job.wpp = w;
  }
  { std::lock_guard<std::mutex> lk(mu_); cur_task_->jobs.push_back(std::move(job)); }
  cv_work_.notify_all();
}

// ------------------------------------------------------------------------------------------------ frame-parallel parsing
void Decoder::set_threads(int n) {
  if (n < 1) n = 1;
  if (n > 16) n = 16;
  if (!workers_.empty() || cur_ || !pool_.empty()) return;          // only before decoding starts
  threads_ = n;
  if (n == 1) return;
  hooks_.self = this;
  hooks_.wait_rows = &Decoder::hook_wait_rows;
  hooks_.rows_done = &Decoder::hook_rows_done;
  for (int i = 0; i < n; i++) workers_.emplace_back([this] { worker_main(); });
  for (int i = 0; i < std::min(6, std::max(3, n)); i++) hash_threads_.emplace_back([this] { hash_main(); });
}

void Decoder::hook_wait_rows(void* self, const PicData* pic, int rows) {
  Decoder* d = static_cast<Decoder*>(self);
  std::unique_lock<std::mutex> lk(d->mu_);
  d->cv_progress_.wait(lk, [&] { return d->stop_ || pic->rows_done.load(std::memory_order_acquire) >= rows; });
}

void Decoder::hook_rows_done(void* self, PicData* pic, int rows) {
  Decoder* d = static_cast<Decoder*>(self);
  { std::lock_guard<std::mutex> lk(d->mu_); pic->rows_done.store(rows, std::memory_order_release); }
  d->cv_progress_.notify_all();
}

// a parser thread: takes a picture and parses its slice segments in the order they arrive, until the picture is closed
void Decoder::worker_main() {
  for (;;) {
    PicTask* t = nullptr;
    WppSession help;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_work_.wait(lk, [&] { return stop_ || !runnable_.empty() || !wpp_sessions_.empty(); });
      if (stop_) return;
      if (!wpp_sessions_.empty()) help = wpp_sessions_.front();        // rows of a picture in progress come before a new picture
      else { t = runnable_.front(); runnable_.pop_front(); }
    }
    if (help.job) { run_wpp_rows(*help.job, help.task); continue; }
    for (;;) {
      SliceJob job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [&] { return stop_ || !t->jobs.empty() || t->closed; });
        if (stop_) return;
        if (t->jobs.empty()) break;
        job = std::move(t->jobs.front());
        t->jobs.pop_front();
      }
      if (!t->error.empty()) continue;                     // the picture is already lost: drain its jobs
      if (job.wpp) {
        // open the rows to the other parser threads, take part, and wait until the last row is done
        auto shared_job = std::make_shared<SliceJob>(job);
        { std::lock_guard<std::mutex> lk(mu_); { WppSession ses; ses.job = shared_job; ses.task = t; wpp_sessions_.push_back(ses); } }
        cv_work_.notify_all();
        run_wpp_rows(*shared_job, t);
        {
          std::unique_lock<std::mutex> lk(shared_job->wpp->mu);
          shared_job->wpp->cv.wait(lk, [&] { return shared_job->wpp->rows_finished.load() >= shared_job->wpp->rows; });
        }
        {
          std::lock_guard<std::mutex> lk(mu_);
          if (shared_job->wpp->failed.load()) t->error = shared_job->wpp->error.empty() ? "wavefront row could not be parsed" : shared_job->wpp->error;
          else t->state.next_ctb_ts = t->pic->num_ctbs;
        }
        continue;
      }
      try {
        SliceDecoder sd(*job.sps, *job.pps, *t->pic, t->state, &hooks_);
        sd.decode(job.sh, job.slice_idx, job.rbsp.data(), job.rbsp.size());
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(mu_);
        t->error = e.what();
      }
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      t->pic->rows_done.store(t->pic->ctbs_h, std::memory_order_release);
      t->pic->parse_done.store(true, std::memory_order_release);
    }
    cv_progress_.notify_all();
  }
}

void Decoder::run_wpp_rows(const SliceJob& job, PicTask* t) {
  WppShared& w = *job.wpp;
  for (;;) {
    const int row = w.next_row.fetch_add(1);
    if (row >= w.rows) break;
    if (row == w.rows - 1) {                               // the last row is claimed: nothing left to offer to other threads
      std::lock_guard<std::mutex> lk(mu_);
      for (auto it = wpp_sessions_.begin(); it != wpp_sessions_.end(); ++it)
        if (it->job->wpp.get() == &w) { wpp_sessions_.erase(it); break; }
    }
    PicParseState st;
    SliceDecoder sd(*job.sps, *job.pps, *t->pic, st, &hooks_);
    if (w.tiles) sd.decode_tile(job.sh, job.slice_idx, job.rbsp.data(), job.rbsp.size(), row, w);
    else sd.decode_wpp_row(job.sh, job.slice_idx, job.rbsp.data(), job.rbsp.size(), row, w);
    { std::lock_guard<std::mutex> lk(w.mu); w.rows_finished.fetch_add(1); }
    w.cv.notify_all();
  }
}

void Decoder::close_current() {
  if (cur_task_) {
    { std::lock_guard<std::mutex> lk(mu_); cur_task_->closed = true; }
    cv_work_.notify_all();
  } else if (cur_) {                       // a picture that never got a parser thread (its first slice failed): dropped
    cur_->is_reference = cur_->is_long_term = cur_->needed_for_output = cur_->in_flight = false;
  }
  cur_task_ = nullptr;
  cur_ = nullptr;
}

// pictures whose parsing is complete leave the pipeline in decoding order: device work, hash check, output decision
void Decoder::retire_ready(bool wait_all) {
  while (!inflight_.empty()) {
    PicTask* t = inflight_.front().get();
    if (t == cur_task_) break;                             // still open: more slices may come
    if (!t->pic->parse_done.load(std::memory_order_acquire)) {
      if (!wait_all) break;
      std::unique_lock<std::mutex> lk(mu_);
      cv_progress_.wait(lk, [&] { return t->pic->parse_done.load(std::memory_order_acquire); });
    }
    PicData* p = t->pic;
    const std::string error = t->error;
    const int parsed = t->state.next_ctb_ts;
    for (PicData* r : t->held) r->users.fetch_sub(1);
    p->in_flight = false;
    inflight_.pop_front();
    if (!error.empty()) {                                  // the slice data could not be parsed: the picture is dropped
      p->is_reference = p->is_long_term = false;
      p->needed_for_output = false;
      p->decoded = false;
      // reported once the unit that is being pushed has been dealt with (push()): the error belongs to an EARLIER picture, and
      // the pictures behind it in the pipeline still have to be retired
      if (deferred_error_.empty()) deferred_error_ = error;
      continue;
    }
    submit_picture(p, parsed);
    begin_output_scan(max_tl_);
    while (PicData* o = next_output(false)) out_queue_.push_back(o);
  }
  flush_batch();
}

void Decoder::queue_flush() {
  retire_ready(true);
  begin_output_scan(max_tl_);
  while (PicData* o = next_output(true)) out_queue_.push_back(o);
  last_display_poc = -(1 << 30);
}

PicData* Decoder::pop_output() {
  if (out_queue_.empty()) return nullptr;
  PicData* p = out_queue_.front();
  out_queue_.pop_front();
  p->lent = true;
  return p;
}

void Decoder::finish_picture() {
  if (threaded()) {                          // close the open picture and wait for everything in flight
    close_current();
    retire_ready(true);
    return;
  }
  if (!cur_) return;
  PicData* p = cur_;
  cur_ = nullptr;
  submit_picture(p, parse_state_.next_ctb_ts);
  flush_batch();
}

void Decoder::submit_picture(PicData* p, int parsed_ctbs) {
  if (p->slices.empty()) {                 // no slice of the picture could be decoded: there is nothing to reconstruct or to show
    p->is_reference = p->is_long_term = p->needed_for_output = false;
    p->decoded = false;
    p->rows_done.store(p->ctbs_h, std::memory_order_release);
    fprintf(stderr, "hmdec: POC %d dropped (no decodable slice)\n", p->poc);
    return;
  }
  if (parsed_ctbs < p->num_ctbs) {
    // slices were lost: HM conceals nothing either (TDecTop.cpp:560 "Warning: ... lost"); the missing CTUs stay as they are
    fprintf(stderr, "hmdec: POC %d is incomplete (%d of %d CTUs)\n", p->poc, parsed_ctbs, p->num_ctbs);
  }
  p->rows_done.store(p->ctbs_h, std::memory_order_release);
  for (int rs = 0; rs < p->num_ctbs; rs++) if (p->slice_addr[rs] < 0) p->reset_ctu(rs);     // CTUs no slice delivered
  if (gpu_) {
    // the picture joins the pictures retired with it if it predicts from none of them (the B pictures of one temporal level, pictures of
    // different sub-GOPs): one set of launches for all of them (hmgpu_decompress_pictures, TDecGop.cpp:105 / TDecTop.cpp:672 per picture)
    bool dependent = batch_.size() >= 16;
    for (auto& sl : p->slices)
      for (int l = 0; l < 2; l++)
        for (int r = 0; r < 16; r++)
          for (PicData* b : batch_) if (sl->ref_pics[l][r] == b) dependent = true;
    if (dependent) flush_batch();
    batch_.push_back(p);
  }
  p->decoded = true;
  p->filtered = true;
  p->needed_for_output = p->pic_output;
  last_decoded_ = p;
  pictures_decoded_++;
  if (!gpu_ && check_hash_ && p->sei_hash_method) check_hash(p);
}

// the device work of the pictures retired together: reconstruction and loop filters of all of them in one batch of launches each
void Decoder::flush_batch() {
  if (batch_.empty() || !gpu_) { batch_.clear(); return; }
  const size_t n = batch_.size();
  std::vector<hmgpu_ctu_meta> metas(n);
  std::vector<hmgpu_coeffs> coefs(n);
  std::vector<std::vector<const hmgpu_slice_params*>> slices(n);
  std::vector<hmgpu_picture_job> jobs(n);
  std::vector<hmgpu_pic_params> pps(n);
  std::vector<hmgpu_filter_job> fjobs(n);
  bool resync = false;
  for (size_t i = 0; i < n; i++) {
    PicData* p = batch_[i];
    hmgpu_ctu_meta& m = metas[i];
    memset(&m, 0, sizeof(m));
    m.depth = p->depth.data(); m.part_size = p->part_size.data(); m.pred_mode = p->pred_mode.data(); m.qp = p->qp.data();
    m.tr_idx = p->tr_idx.data();
    for (int c = 0; c < 3; c++) { m.cbf[c] = p->cbf[c].data(); m.transform_skip[c] = p->ts[c].data(); }
    for (int l = 0; l < 2; l++) { m.mv[l] = p->mv[l].data(); m.ref_idx[l] = p->ref_idx[l].data(); }
    m.intra_dir[0] = p->intra_dir[0].data(); m.intra_dir[1] = p->intra_dir[1].data();
    m.transquant_bypass = p->has_bypass ? p->bypass.data() : nullptr;
    m.ipcm = p->has_pcm ? p->ipcm.data() : nullptr;
    m.slice_idx = p->slice_idx.data();
    m.tile_idx = p->tile_idx.data();
    for (int k = 0; k < 2; k++) m.ccp_alpha[k] = p->ccp[k].empty() ? nullptr : p->ccp[k].data();      // 4:4:4: cross-component prediction weights
    hmgpu_coeffs& co = coefs[i];
    memset(&co, 0, sizeof(co));
    for (int c = 0; c < 3; c++) { co.level[c] = p->coeff[c].data(); co.pcm_sample[c] = p->has_pcm ? p->pcm[c].data() : nullptr; }
    if (p->compact) {
      // CTUs nobody parsed hold no TUs: their start is the next CTU's; the last entry is the length of the stream
      bool in_order = true;
      for (int c = 0; c < 3; c++) {
        p->level_start[c][p->num_ctbs] = p->level_cursor[c];
        for (int rs = p->num_ctbs - 1; rs >= 0; rs--) if (p->slice_addr[rs] < 0) p->level_start[c][rs] = p->level_start[c][rs + 1];
        const uint32_t per = (uint32_t)(1u << (2 * p->log2_ctb)) >> (c ? 2 : 0);
        for (int rs = 0; rs < p->num_ctbs; rs++)
          in_order &= p->level_start[c][rs + 1] >= p->level_start[c][rs] && p->level_start[c][rs + 1] - p->level_start[c][rs] <= per;
        co.ctu_level_start[c] = p->level_start[c].data();
      }
      if (!in_order) {
        // a damaged stream delivered CTUs twice or out of order: their pieces do not follow each other.  HM's dense layout, rebuilt from
        // the per-TU offsets, for this one picture (ordinary arrays: copied before the call below returns to the parser)
        p->expand_dense();
        for (int c = 0; c < 3; c++) { co.level[c] = p->dense_levels[c].data(); co.ctu_level_start[c] = nullptr; }
        resync = true;
      }
    }
    for (auto& s : p->slices) slices[i].push_back(&s->params);
    jobs[i].pic = p->handle; jobs[i].num_slices = (int32_t)slices[i].size(); jobs[i].slices = slices[i].data();
    jobs[i].meta = &m; jobs[i].coeffs = &co;
    memset(&pps[i], 0, sizeof(pps[i]));
    pps[i].lf_across_tiles = p->lf_across_tiles;
    pps[i].sao_enabled = p->sao_enabled;
    pps[i].sao_offset_shift_luma = p->sao_offset_shift[0]; pps[i].sao_offset_shift_chroma = p->sao_offset_shift[1];
    fjobs[i].pic = p->handle; fjobs[i].pp = &pps[i]; fjobs[i].sao = p->sao_enabled ? p->sao.data() : nullptr;
  }
  std::vector<PicData*> pics;
  pics.swap(batch_);
  // placement: the pictures of the batch round-robin over the contexts (a batch of one stays where its first reference lives: a chain of
  // P pictures does not hop); every reference picture a context has not seen yet is copied there once, behind its reconstruction
  const size_t nd = gpus_.size();
  std::vector<int> where(n, 0);
  if (nd > 1) {
    for (size_t i = 0; i < n; i++) where[i] = (int)((rr_ + i) % nd);
    static const bool hop = getenv("HMDEC_PLACE_ROUND_ROBIN") != nullptr && getenv("HMDEC_PLACE_ROUND_ROBIN")[0] == '1';   // (tests: every picture moves on)
    if (n == 1 && !hop) {
      const PicData* r = nullptr;
      for (auto& sl : pics[0]->slices) for (int l = 0; l < 2 && !r; l++) for (int k = 0; k < 16 && !r; k++) r = sl->ref_pics[l][k];
      if (r) where[0] = r->home;
    } else {
      rr_ += n;
    }
  }
  for (size_t d = 0; d < nd; d++) {
    std::vector<hmgpu_picture_job> dj;
    std::vector<hmgpu_filter_job> df;
    for (size_t i = 0; i < n; i++) {
      if (where[i] != (int)d) continue;
      for (auto& sl : pics[i]->slices)
        for (int l = 0; l < 2; l++)
          for (int k = 0; k < 16; k++) {
            PicData* r = sl->ref_pics[l][k];
            if (!r || ((r->present >> d) & 1u)) continue;
            const hmgpu_status ts = hmgpu_picture_transfer(gpus_[r->home], r->handle, gpus_[d], r->handle);
            if (ts != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_picture_transfer: ") + hmgpu_status_string(ts));
            r->present |= 1u << d;
          }
      dj.push_back(jobs[i]);
      df.push_back(fjobs[i]);
    }
    if (dj.empty()) continue;
    hmgpu_status st = hmgpu_decompress_pictures(gpus_[d], (int32_t)dj.size(), dj.data());
    if (st != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_decompress_pictures: ") + hmgpu_status_string(st));
    if (resync) (void)hmgpu_sync(gpus_[d]);                  // (the dense stand-in of a damaged picture is not page-locked staging: let its copies finish)
    st = hmgpu_filter_pictures(gpus_[d], (int32_t)df.size(), df.data());
    if (st != HMGPU_OK) throw std::runtime_error(std::string("hmgpu_filter_pictures: ") + hmgpu_status_string(st));
  }
  for (size_t i = 0; i < n; i++) { pics[i]->home = where[i]; pics[i]->present = 1u << where[i]; }
  ++submitted_seq_;
  batches_submitted_++;
  for (PicData* p : pics) p->submit_seq = submitted_seq_;
  if (check_hash_) for (PicData* p : pics) if (p->sei_hash_method) check_hash(p);
}

bool Decoder::fetch_planes(PicData* pic) {
  if (!gpu_ || !pic) return false;
  if (pic->planes_valid) return true;
  flush_batch();
  if (const uint64_t t = pic->dl_ticket.load()) {        // begun for the hash check: wait for it instead of copying again
    if (hmgpu_download_wait(ctx_of(pic), t) != HMGPU_OK) return false;
    pic->planes_valid = true;
    return true;
  }
  pic->plane[0].resize((size_t)pic->width * pic->height);
  pic->plane[1].resize(((size_t)pic->width * pic->height) >> pic->cshift);
  pic->plane[2].resize(((size_t)pic->width * pic->height) >> pic->cshift);
  int16_t* planes[3] = {pic->plane[0].data(), pic->plane[1].data(), pic->plane[2].data()};
  const int32_t strides[3] = {pic->width, pic->width >> pic->csx, pic->width >> pic->csx};
  const uint64_t seq = submitted_seq_;
  if (hmgpu_picture_download(ctx_of(pic), pic->handle, planes, strides) != HMGPU_OK) return false;
  if (gpus_.size() == 1) synced_seq_ = seq;  // a download returns after everything enqueued before it (on that context)
  pic->planes_valid = true;
  return true;
}

// MD5 of one plane as the SEI defines it (TComPicYuvMD5.cpp:183-205): samples as 1 or 2 little-endian bytes, row by row
bool Decoder::md5_plane_matches(const PicData* pic, int comp, int bd, const uint8_t want[16]) {
  Md5 md5;
  const HostVec<int16_t>& pl = pic->plane[comp];
  if (bd > 8) {
    md5.update(reinterpret_cast<const uint8_t*>(pl.data()), pl.size() * 2);        // int16 samples on a little-endian host: the bytes as they lie
  } else {
    uint8_t buf[4096];
    for (size_t i = 0; i < pl.size(); i += sizeof(buf)) {
      const size_t n = std::min(sizeof(buf), pl.size() - i);
      for (size_t k = 0; k < n; k++) buf[k] = (uint8_t)pl[i + k];
      md5.update(buf, n);
    }
  }
  uint8_t got[16];
  md5.final(got);
  return memcmp(got, want, 16) == 0;
}

void Decoder::hash_main() {
  for (;;) {
    HashJob j;
    {
      std::unique_lock<std::mutex> lk(hash_mu_);
      hash_cv_.wait(lk, [&] { return hash_stop_ || !hash_jobs_.empty(); });
      if (hash_jobs_.empty()) return;
      j = hash_jobs_.front();
      hash_jobs_.pop_front();
      hash_busy_++;
    }
    bool landed = true;
    if (const uint64_t t = j.pic->dl_ticket.load()) landed = hmgpu_download_wait(ctx_of(j.pic), t) == HMGPU_OK;
    if (!landed || !md5_plane_matches(j.pic, j.comp, j.bd, j.want)) {
      if (!j.pic->hash_mismatch.exchange(true)) {
        hash_mismatches_++;
        fprintf(stderr, landed ? "hmdec: ***ERROR*** decoded picture hash mismatch, POC %d\n" : "hmdec: ***ERROR*** device error behind the picture of POC %d (not checked)\n", j.pic->poc);
      }
    }
    j.pic->users.fetch_sub(1);
    {
      std::lock_guard<std::mutex> lk(hash_mu_);
      hash_busy_--;
    }
    hash_idle_cv_.notify_all();
    cv_progress_.notify_all();             // (a picture buffer may have become free)
  }
}

// device MD5: the oldest outstanding digests, as far as they are there (all of them: block)
void Decoder::poll_device_hashes(bool block) {
  while (gpu_ && !dev_hashes_.empty()) {
    uint8_t got[3][16];
    int32_t len = 0, ready = 0;
    if (hmgpu_hash_wait(dev_hashes_.front().ctx, dev_hashes_.front().ticket, block ? 1 : 0, got, &len, &ready) != HMGPU_OK) { ready = 1; len = 0; }
    if (!ready) return;
    const DevHash& h = dev_hashes_.front();
    if (len != 16 || memcmp(got, h.want, 16 * h.ncomp) != 0) {
      hash_mismatches_++;
      fprintf(stderr, "hmdec: ***ERROR*** decoded picture hash mismatch, POC %d\n", h.poc);
    }
    dev_hashes_.pop_front();
  }
}

void Decoder::drain_hash_jobs() {
  poll_device_hashes(true);
  if (hash_threads_.empty()) return;
  std::unique_lock<std::mutex> lk(hash_mu_);
  hash_idle_cv_.wait(lk, [&] { return hash_jobs_.empty() && hash_busy_ == 0; });
}

// TDecGop.cpp:199-262: the reconstruction against the decoded picture hash SEI
void Decoder::check_hash(PicData* pic) {
  if (!gpu_) return;
  uint8_t got[3][16];
  memset(got, 0, sizeof(got));
  int len = 0;
  if (pic->sei_hash_method == 1 && device_md5_) {
    // the chains run on the device (hmgpu_picture_hash_begin): no download, no hash threads, the picture buffer is free at once;
    // the verdict arrives a fraction of a second later (a GPU lane runs the serial chain ~8x slower than a host core) and is read
    // when it is there, at the latest when somebody asks (hash_mismatches, flush, end of the sequence)
    while (dev_hashes_.size() >= 64) poll_device_hashes(true);         // (the device keeps a ring of 96)
    DevHash h;
    h.poc = pic->poc;
    h.ncomp = pic->num_comps;
    memcpy(h.want, pic->sei_hash, sizeof(h.want));
    h.ctx = ctx_of(pic);
    if (hmgpu_picture_hash_begin(h.ctx, pic->handle, 1, &h.ticket) != HMGPU_OK) return;
    dev_hashes_.push_back(h);
    poll_device_hashes(false);
    return;
  }
  if (pic->sei_hash_method == 1) {
    // MD5 is a serial chain over every byte of a plane (~20-30 ms for a 2160p luma plane): with parser threads the three planes
    // go to the hash threads, picture after picture, and the decoding thread carries on; the picture stays pinned meanwhile
    if (!hash_threads_.empty()) {
      // the copy is only enqueued here: the hash threads wait for it, the decoding thread goes on submitting pictures
      if (!pic->planes_valid && !pic->dl_ticket.load()) {
        pic->plane[0].resize((size_t)pic->width * pic->height);
        pic->plane[1].resize(((size_t)pic->width * pic->height) >> pic->cshift);
        pic->plane[2].resize(((size_t)pic->width * pic->height) >> pic->cshift);
        int16_t* planes[3] = {pic->plane[0].data(), pic->plane[1].data(), pic->plane[2].data()};
        const int32_t strides[3] = {pic->width, pic->width >> pic->csx, pic->width >> pic->csx};
        uint64_t t = 0;
        if (hmgpu_picture_download_begin(ctx_of(pic), pic->handle, planes, strides, &t) != HMGPU_OK) return;
        pic->dl_ticket.store(t);
      }
      pic->users.fetch_add(pic->num_comps);
      {
        std::lock_guard<std::mutex> lk(hash_mu_);
        for (int c = 0; c < pic->num_comps; c++) {
          HashJob j;
          j.pic = pic; j.comp = c; j.bd = c ? sps_->bit_depth_chroma : sps_->bit_depth_luma;
          memcpy(j.want, pic->sei_hash[c], 16);
          hash_jobs_.push_back(j);
        }
      }
      hash_cv_.notify_all();
      return;
    }
    if (!fetch_planes(pic)) return;
    bool ok = true;
    for (int c = 0; c < pic->num_comps; c++) ok &= md5_plane_matches(pic, c, c ? sps_->bit_depth_chroma : sps_->bit_depth_luma, pic->sei_hash[c]);
    if (!ok) {
      pic->hash_mismatch = true;
      hash_mismatches_++;
      fprintf(stderr, "hmdec: ***ERROR*** decoded picture hash mismatch, POC %d\n", pic->poc);
    }
    return;
  } else {
    int32_t n = 0;
    if (hmgpu_picture_hash(ctx_of(pic), pic->handle, pic->sei_hash_method, got, &n) != HMGPU_OK) return;
    len = n;
  }
  bool ok = true;
  for (int c = 0; c < pic->num_comps; c++) if (memcmp(got[c], pic->sei_hash[c], len)) ok = false;
  if (!ok) {
    pic->hash_mismatch = true;
    hash_mismatches_++;
    fprintf(stderr, "hmdec: ***ERROR*** decoded picture hash mismatch, POC %d\n", pic->poc);
  }
}

void Decoder::flush() { finish_picture(); }

// ------------------------------------------------------------------------------------------------ output (libHM's rules)
void Decoder::begin_output_scan(int max_tl) {
  scan_.clear();
  for (auto& p : pool_) if (p->decoded && p.get() != cur_) scan_.push_back(p.get());
  std::sort(scan_.begin(), scan_.end(), [](const PicData* a, const PicData* b) { return a->poc < b->poc; });
  scan_idx_ = 0;
  num_not_displayed_ = dpb_fullness_ = 0;
  if (!sps_) return;
  const int layers = sps_->max_sub_layers;
  const int t = (max_tl == -1 || max_tl >= layers) ? layers - 1 : max_tl;
  num_reorder_ = sps_->num_reorder_pics[t];
  max_dec_buffering_ = sps_->max_dec_pic_buffering[t];            // HM stores sps_max_dec_pic_buffering_minus1 + 1 too (TDecCAVLC.cpp parseSPS)
  for (PicData* p : scan_) {
    if (p->needed_for_output && p->poc > last_display_poc) { num_not_displayed_++; dpb_fullness_++; }
    else if (p->is_reference) dpb_fullness_++;
  }
}

PicData* Decoder::next_output(bool flush_all) {
  while (scan_idx_ < scan_.size()) {
    PicData* p = scan_[scan_idx_];
    if ((flush_all && p->needed_for_output) ||
        (p->needed_for_output && p->poc > last_display_poc && (num_not_displayed_ > num_reorder_ || dpb_fullness_ > max_dec_buffering_))) {
      if (!flush_all) num_not_displayed_--;
      if (!p->is_reference) dpb_fullness_--;
      last_display_poc = p->poc;
      p->needed_for_output = false;
      p->lent = true;
      return p;
    }
    scan_idx_++;
  }
  return nullptr;
}

}  // namespace hmdec
