// decoder.h -- the host half of an HEVC decoder around the device path: NAL units in, finished pictures out.
// Parsing (parameter sets, slice headers, CABAC slice data), picture order count (8.3.1), reference picture sets (8.3.2), reference
// picture lists (8.3.4) and the output order stay on the host; every sample is produced by libhmgpu through the two drop-in calls.
// HM counterpart: TDecTop.cpp (decode / xActivateParameterSets / xDecodeSlice / executeLoopFilters), TComSlice.cpp (setRefPicList,
// applyReferencePictureSet), TDecGop.cpp, SEIread.cpp (decoded picture hash).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "picture.h"
#include "slice_decoder.h"

namespace hmdec {

class Decoder {
 public:
  Decoder();
  ~Decoder();
  // configuration (before the first NAL unit)
  void set_parse_only(bool v) { parse_only_ = v; }        // no device: metadata only (host-side tests)
  void set_check_hash(bool v) { check_hash_ = v; }
  void set_device(int ordinal) { device_ = ordinal; devices_.assign(1, ordinal); }
  // Several device contexts (one per GPU; the same ordinal twice gives two contexts on one GPU): the pictures retired together -- the B
  // pictures of one temporal level, TDecTop.cpp:672 per picture -- are placed round-robin on them, each reference picture is copied once to
  // every context that predicts from it (hmgpu_picture_transfer), the output order is HM's as ever.
  void set_devices(const int* ordinals, int n) { devices_.assign(ordinals, ordinals + (n > 0 ? n : 0)); if (devices_.empty()) devices_.assign(1, device_); device_ = devices_[0]; }
  int num_devices() const { return (int)devices_.size(); }
  uint64_t transfer_bytes() const;                        // bytes of reference pictures copied between the contexts so far
  // Parser threads (frame-parallel parsing): 1 = everything on the caller's thread (default).  With n > 1 the slice data of up to n
  // pictures is parsed concurrently, a picture running at most one CTB row behind the picture it takes temporal motion vectors
  // from; pictures are handed to the device, checked and put out in decoding order all the same, only later.
  void set_threads(int n);
  bool threaded() const { return threads_ > 1; }
  void retire_ready(bool wait_all);                       // threaded: finished pictures go to the device and to the output queue
  void queue_flush();                                     // threaded: everything decoded and still waiting becomes output
  PicData* pop_output();
  // One NAL unit (with or without a start code).  Returns true when the unit starts a new picture while the previous one was still
  // open: that picture has then been finished and the SAME unit must be pushed again (libHM's bNewPicture protocol).
  bool push(const uint8_t* data, size_t len, int max_temporal_layer, int* nal_type_out);
  std::string take_deferred_error() { std::string e; e.swap(deferred_error_); return e; }   // error of a picture retired earlier, if any
  void finish_picture();                                  // TDecTop::executeLoopFilters
  void flush();                                           // end of stream / EOS: everything still waiting becomes output
  // output: libHM's rules (libHMDecoder.cpp:248-339)
  void begin_output_scan(int max_temporal_layer);
  PicData* next_output(bool flush_all);
  const Sps* active_sps() const { return sps_.get(); }
  PicData* last_decoded() const { return last_decoded_; }
  PicData* open_picture() const { return cur_; }
  int hash_mismatches() { drain_hash_jobs(); return hash_mismatches_.load(); }   // (waits for the MD5 checks still running on the hash threads)
  int pictures_decoded() const { return pictures_decoded_; }
  int device_batches() const { return (int)batches_submitted_; }
  void set_device_md5(bool on) { device_md5_ = on; }        // MD5 hash SEIs are checked on the device instead of on the hash threads
  const std::string& last_error() const { return last_error_; }
  void set_error(const std::string& s) { last_error_ = s; }
  bool fetch_planes(PicData* pic);                        // device -> host planes of a finished picture (no-op when parse-only)
  int last_display_poc = -(1 << 30);

 private:
  void activate(const SliceHeader& sh);
  bool opens_new_sequence(const SliceHeader& sh) const;
  void start_picture(const SliceHeader& sh);
  void decode_slice(const std::vector<uint8_t>& rbsp, BitReader& br, SliceHeader& sh, const std::vector<size_t>& epb);
  int compute_poc(const SliceHeader& sh);
  void apply_rps(const SliceHeader& sh);
  void build_ref_lists(const SliceHeader& sh, SliceInfo& si);
  void build_slice_params(const SliceHeader& sh, SliceInfo& si);
  void parse_sei(const std::vector<uint8_t>& rbsp, bool suffix);
  void check_hash(PicData* pic);
  void poll_device_hashes(bool block);
  void submit_picture(PicData* pic, int parsed_ctbs);     // marks of a completely parsed picture; its device work joins the batch
  void flush_batch();                                     // the device work of the pictures retired together (hmgpu_decompress_pictures / hmgpu_filter_pictures)
  void close_current();
  void worker_main();
  static void hook_wait_rows(void* self, const PicData* pic, int rows);
  static void hook_rows_done(void* self, PicData* pic, int rows);
  PicData* acquire_buffer();
  PicData* find_ref(int poc, bool lsb_only, bool any_marking);

  ParamSets ps_;
  std::shared_ptr<Sps> sps_;
  std::shared_ptr<Pps> pps_;
  ZScan zscan_;
  hmgpu_ctx* gpu_ = nullptr;                               // the first context (owner of the staging blocks); null: parse only
  std::vector<hmgpu_ctx*> gpus_;                           // all of them (set_devices), gpus_[0] == gpu_
  std::vector<int> devices_{0};
  hmgpu_ctx* ctx_of(const PicData* p) const { return gpus_.empty() ? gpu_ : gpus_[p->home]; }
  void sync_all();
  uint64_t rr_ = 0;                                        // round-robin position of the next batch
  uint64_t transfer_bytes_closed_ = 0;                     // ... of the contexts of earlier sequences
  hmgpu_seq_params seq_{};
  std::vector<std::unique_ptr<PicData>> pool_;            // DPB + free buffers
  std::vector<std::unique_ptr<PicData>> retired_;         // pictures of the previous sequence the application may still hold
  PicData* cur_ = nullptr;
  PicData* last_decoded_ = nullptr;
  PicParseState parse_state_;
  SliceHeader last_independent_;
  bool have_independent_ = false;
  // 8.3.1 / 8.3.2 state
  int prev_tid0_poc_ = 0;
  bool first_picture_ = true, after_eos_ = false, no_rasl_output_ = false;
  int poc_cra_ = 0;
  bool skip_rasl_ = false;
  std::vector<PicData*> st_before_, st_after_, lt_curr_;
  // pending SEI
  bool pending_hash_ = false;
  int pending_hash_method_ = 0;
  uint8_t pending_hash_val_[3][16];
  // output scan state
  int num_not_displayed_ = 0, dpb_fullness_ = 0, num_reorder_ = 0, max_dec_buffering_ = 0;
  size_t scan_idx_ = 0;
  std::vector<PicData*> scan_;
  bool parse_only_ = false, check_hash_ = true;
  int device_ = 0, pictures_decoded_ = 0;
  std::atomic<int> hash_mismatches_{0};
  // MD5 checks of the decoded-picture-hash SEI off the decoding thread: one job per plane, a few threads (with parser threads only)
  struct HashJob { PicData* pic; int comp, bd; uint8_t want[16]; };
  std::vector<std::thread> hash_threads_;
  std::deque<HashJob> hash_jobs_;
  std::mutex hash_mu_;
  std::condition_variable hash_cv_, hash_idle_cv_;
  int hash_busy_ = 0;
  bool hash_stop_ = false;
  void hash_main();
  void drain_hash_jobs();
  static bool md5_plane_matches(const PicData* pic, int comp, int bd, const uint8_t want[16]);
  uint64_t submitted_seq_ = 0, synced_seq_ = 0;            // device submissions / the last one known to have completed
  struct DevHash { uint64_t ticket; int poc; int ncomp; uint8_t want[3][16]; hmgpu_ctx* ctx; };
  std::deque<DevHash> dev_hashes_;                         // MD5 chains under way on the device (hmgpu_picture_hash_begin)
  bool device_md5_ = !(getenv("HMDEC_DEVICE_MD5") != nullptr && getenv("HMDEC_DEVICE_MD5")[0] == '0');   // default on
  std::vector<PicData*> batch_;                            // pictures retired and not yet submitted: mutually independent
  uint64_t batches_submitted_ = 0;
  std::string last_error_;
  std::string deferred_error_;                            // parse error of a picture that left the pipeline while another unit was pushed
  bool push_unit(const uint8_t* data, size_t len, int max_temporal_layer, int* nal_type_out);
  // ---- frame-parallel parsing
  struct SliceJob {
    SliceHeader sh; int slice_idx = 0; std::vector<uint8_t> rbsp; std::shared_ptr<Sps> sps; std::shared_ptr<Pps> pps;
    std::shared_ptr<WppShared> wpp;                       // set: the segment is a whole wavefront-coded picture, parsed row-parallel
  };

  struct PicTask {
    PicData* pic = nullptr;
    std::deque<SliceJob> jobs;
    bool closed = false;                                  // no more slices will come
    PicParseState state;
    std::vector<PicData*> held;                           // reference pictures kept alive for the parser thread
    std::string error;
  };
  struct WppSession { std::shared_ptr<SliceJob> job; PicTask* task = nullptr; };
  void run_wpp_rows(const SliceJob& job, PicTask* t);     // claims and parses rows of a wavefront picture until none is left
  int threads_ = 1, max_tl_ = -1;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_progress_;
  std::deque<std::unique_ptr<PicTask>> inflight_;         // decoding order
  std::deque<PicTask*> runnable_;
  std::deque<WppSession> wpp_sessions_;                    // wavefront pictures with rows nobody has claimed yet
  PicTask* cur_task_ = nullptr;
  bool stop_ = false;
  ProgressHooks hooks_;
  std::deque<PicData*> out_queue_;
  // a unit that opened a new picture has been taken in already (threaded mode): its repetition by the caller is recognised and skipped
  bool repush_pending_ = false;
  size_t repush_len_ = 0;
  uint8_t repush_head_[24] = {0};
};

}  // namespace hmdec
