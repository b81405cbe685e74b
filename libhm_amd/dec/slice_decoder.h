// slice_decoder.h -- slice_segment_data() of Rec. ITU-T H.265 7.3.8 parsed into HM's per-CTU arrays (picture.h), together with the
// decoding processes that feed on parsed data only: QP derivation (8.6.1), intra mode derivation (8.4.2, 8.4.3) and motion vector
// prediction (8.5.3.2: merge, AMVP, temporal candidates).  HM counterpart: TDecSlice.cpp, TDecCu.cpp:142-372, TDecEntropy.cpp,
// TDecSbac.cpp, TComDataCU.cpp (getInterMergeCandidates :2400-2710, fillMvpCand :2790-3000, xGetColMVP :3240-3330).
#pragma once
#include <condition_variable>
#include <mutex>
#include <string>

#include "cabac.h"
#include "picture.h"

namespace hmdec {

// how a parser thread learns that rows of another picture are final (set by the decoder; null = never wait)
struct ProgressHooks {
  void (*wait_rows)(void* self, const PicData* pic, int rows) = nullptr;
  void (*rows_done)(void* self, PicData* pic, int rows) = nullptr;
  void* self = nullptr;
};

struct PicParseState {               // carried from one slice segment of a picture to the next
  ContextSet end_of_segment;         // 9.3.2.4 storage for dependent slice segments
  bool have_end_of_segment = false;
  int last_qp = 0;                   // qPY_PREV
  int next_ctb_ts = 0;
  ContextSet wpp;                    // 9.3.2.4 storage after the second CTB of a row (entropy_coding_sync)
  bool wpp_valid = false;
};

// Wavefront parallel processing inside one picture (entropy_coding_sync_enabled_flag, one slice segment, no tiles): every CTB row
// is a sub-stream of its own that starts from the context variables the row above had after its second CTB, so rows can be parsed
// by different threads, each at least two CTBs behind the row above (9.3.1, 9.3.2.4).  Rows are claimed in increasing order from
// `next_row`; whoever claims a row parses it, so a row only ever waits for rows that are being parsed or are finished.
struct WppShared {
  int rows = 0, width = 0;                         // units: CTB rows of `width` CTBs, or (tiles = true) the picture's tiles
  bool tiles = false;
  std::vector<size_t> row_bit_pos;                 // where the sub-stream of each row starts in the RBSP
  std::vector<ContextSet> ctx_after2;              // context variables after the second CTB of each row
  std::unique_ptr<std::atomic<int>[]> progress;    // CTBs of each row that are finished
  std::atomic<int> next_row{0}, rows_finished{0};
  std::atomic<bool> failed{false};
  std::mutex mu;
  std::condition_variable cv;
  std::string error;
  void init(int r, int w) {
    rows = r; width = w;
    row_bit_pos.assign(r, 0);
    ctx_after2.resize(r);
    progress.reset(new std::atomic<int>[r]);
    for (int i = 0; i < r; i++) progress[i].store(0);
  }
  void publish(int row, int done) {
    { std::lock_guard<std::mutex> lk(mu); progress[row].store(done, std::memory_order_release); }
    cv.notify_all();
  }
  void wait(int row, int need) {
    if (progress[row].load(std::memory_order_acquire) >= need) return;
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return progress[row].load(std::memory_order_acquire) >= need; });
  }
};

class SliceDecoder {
 public:
  SliceDecoder(const Sps& sps, const Pps& pps, PicData& pic, PicParseState& st, const ProgressHooks* hooks = nullptr)
      : sps_(sps), pps_(pps), pic_(pic), st_(st), hooks_(hooks) {}
  // parses one slice segment; `slice` is the entry of pic.slices it belongs to.  Returns true when the picture is complete.
  bool decode(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes);
  // one CTB row of a wavefront-coded picture (see WppShared); every row has its own SliceDecoder and PicParseState
  void decode_wpp_row(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes, int row, WppShared& w);
  // one tile of a picture coded as one slice segment with an entry point per tile: tiles depend on nothing but themselves
  void decode_tile(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes, int tile, WppShared& w);

 private:
  struct Mv { int16_t x = 0, y = 0; bool operator==(const Mv& o) const { return x == o.x && y == o.y; } };
  struct Motion {
    Mv mv[2];
    int8_t ref[2] = {-1, -1};
    bool same(const Motion& o) const {
      if (ref[0] != o.ref[0] || ref[1] != o.ref[1]) return false;
      for (int l = 0; l < 2; l++) if (ref[l] >= 0 && !(mv[l] == o.mv[l])) return false;
      return true;
    }
  };
  bool decode_segment(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes);
  void init_contexts();
  void ctu(int rs);
  void sao_syntax(int rs);
  void coding_quadtree(int x0, int y0, int log2, int depth);
  void coding_unit(int x0, int y0, int log2, int depth);
  void pcm_sample(int x0, int y0, int log2);
  void prediction_unit(int xcb, int ycb, int ncbs, int x0, int y0, int w, int h, int part_idx, int part_mode, int depth, bool skip);
  void intra_modes(int x0, int y0, int log2, bool nxn);
  void transform_tree(int x0, int y0, int xbase, int ybase, int log2, int tr_depth, int blk, int cu_log2, int parent_cbf_cb, int parent_cbf_cr);
  void residual_coding(int x0, int y0, int log2, int c, int sub = 0);
  int chroma_pred_mode(size_t part) const;
  void qp_delta();
  void start_quant_group(int x0, int y0);
  int cu_qp() const;
  // neighbourhood
  bool available(int xc, int yc, int xn, int yn) const;
  bool pu_available(int xcb, int ycb, int ncbs, int xpb, int ypb, int w, int h, int part_idx, int xn, int yn) const;
  Motion motion_at(int x, int y) const;
  void set_motion(int x0, int y0, int w, int h, const Motion& m);
  void merge_candidates(int xcb, int ycb, int ncbs, int xpb, int ypb, int w, int h, int part_idx, int part_mode, int merge_idx, Motion& out);
  Mv amvp(int xcb, int ycb, int ncbs, int xpb, int ypb, int w, int h, int part_idx, int list, int ref_idx, int mvp_flag);
  bool temporal_mv(int xpb, int ypb, int w, int h, int list, int ref_idx, Mv& out) const;
  bool col_mv(int xcol, int ycol, int list, int ref_idx, Mv& out) const;
  static Mv scale_mv(Mv mv, int tb, int td);
  template <class V, class T> void fill_z(V& a, size_t first, int count, T v) { std::fill(a.begin() + first, a.begin() + first + count, v); }

  const Sps& sps_;
  const Pps& pps_;
  PicData& pic_;
  PicParseState& st_;
  const ProgressHooks* hooks_ = nullptr;
  const SliceHeader* sh_ = nullptr;
  SliceInfo* slice_ = nullptr;
  Cabac cabac_;
  ContextSet ctx_;
  int slice_idx_ = 0, ctb_rs_ = 0, ctb_ts_ = 0;
  // quantisation group state (7.3.8.4, 8.6.1)
  bool is_cu_qp_delta_coded_ = false;
  int cu_qp_delta_val_ = 0, qg_pred_ = 0, qg_x_ = 0, qg_y_ = 0;
  bool first_qg_in_unit_ = true;     // first quantisation group of the slice / tile / WPP row
  // current CU
  int cu_pred_mode_ = 0, cu_x_ = 0, cu_y_ = 0, cu_log2_ = 0;
  bool cu_bypass_ = false;
  int intra_luma_[4] = {0}, intra_chroma_ = 0;
  bool no_backward_pred_ = false;
};

}  // namespace hmdec
