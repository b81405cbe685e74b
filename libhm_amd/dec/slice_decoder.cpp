// slice_decoder.cpp -- see slice_decoder.h.  Clause numbers refer to Rec. ITU-T H.265.
#include "slice_decoder.h"

#include <algorithm>
#include <cstdlib>

namespace hmdec {

namespace {
enum { MODE_INTER = HMGPU_MODE_INTER, MODE_INTRA = HMGPU_MODE_INTRA };
enum { PART_2Nx2N = 0, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };
enum { PRED_L0 = 0, PRED_L1 = 1, PRED_BI = 2 };
const int kDmChroma = 36;            // HM's DM_CHROMA_IDX: "same as luma", kept symbolic in m_puhIntraDir[chroma]

// 6.5.3 - 6.5.5: scan orders for sub-block grids / positions inside a 4x4 sub-block; pos = y * blk + x
struct ScanTables {
  uint8_t order[4][3][64];           // [log2 blk][scanIdx][i] -> pos
  uint8_t index[4][3][64];           // inverse
  ScanTables() {
    for (int l = 0; l < 4; l++) {
      const int blk = 1 << l;
      int i = 0, x = 0, y = 0;
      for (;;) {
        while (y >= 0) {
          if (x < blk && y < blk) order[l][0][i++] = (uint8_t)(y * blk + x);
          y--;
          x++;
        }
        y = x;
        x = 0;
        if (i >= blk * blk) break;
      }
      for (int k = 0; k < blk * blk; k++) {
        order[l][1][k] = (uint8_t)k;                                   // horizontal: raster
        order[l][2][k] = (uint8_t)((k % blk) * blk + k / blk);         // vertical: column by column
      }
      for (int s = 0; s < 3; s++) for (int k = 0; k < blk * blk; k++) index[l][s][order[l][s][k]] = (uint8_t)k;
    }
  }
};
const ScanTables kScan;
const uint8_t kSigCtx4x4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};      // 9.3.4.2.5 ctxIdxMap (index = (yC << 2) + xC)
// 9.3.4.2.5, blocks larger than 4x4: sigCtx by the coded flags of the right / below sub-blocks and the position (yP << 2) + xP
const uint8_t kSigCtxPattern[4][16] = {{2, 1, 1, 0, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0},
                                       {2, 2, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0},
                                       {2, 1, 0, 0, 2, 1, 0, 0, 2, 1, 0, 0, 2, 1, 0, 0},
                                       {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2}};

inline int clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
}  // namespace

// ------------------------------------------------------------------------------------------------ slice segment level
void SliceDecoder::init_contexts() {
  int init_type = 0;
  if (sh_->type == SLICE_P) init_type = sh_->cabac_init_flag ? 2 : 1;
  else if (sh_->type == SLICE_B) init_type = sh_->cabac_init_flag ? 1 : 2;
  ctx_.init(init_type, sh_->qp);
}

bool SliceDecoder::decode(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes) {
  const int first_ts = pps_.ctb_rs_to_ts[sh.segment_address];
  ctb_ts_ = first_ts;
  try {
    return decode_segment(sh, slice_idx, rbsp, bytes);
  } catch (...) {
    // a segment that cannot be parsed leaves nothing behind: its CTUs go back to "never decoded", so that a half-written CTU
    // never reaches the device (HM asserts in this situation; here the picture goes on without the segment)
    for (int ts = first_ts; ts <= ctb_ts_ && ts < pic_.num_ctbs; ts++) {
      const int rs = pps_.ctb_ts_to_rs[ts];
      pic_.reset_ctu(rs);
      pic_.slice_addr[rs] = -1;
    }
    throw;
  }
}

bool SliceDecoder::decode_segment(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes) {
  sh_ = &sh;
  slice_idx_ = slice_idx;
  slice_ = pic_.slices[slice_idx].get();
  cabac_.attach(rbsp, bytes);
  no_backward_pred_ = true;
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < sh.num_ref_idx[l]; i++) if (slice_->ref_poc[l][i] > pic_.poc) no_backward_pred_ = false;
  const int W = pic_.ctbs_w;
  ctb_ts_ = pps_.ctb_rs_to_ts[sh.segment_address];
  if (ctb_ts_ != st_.next_ctb_ts) throw ParseError("slice segment does not continue where the previous one ended");
  if (!sh.dependent) st_.last_qp = sh.qp;           // first quantisation group in a slice: qPY_PREV = SliceQpY (8.6.1)
  cabac_.start(sh.data_bit_offset);
  bool first = true, done = false;
  while (!done) {
    ctb_rs_ = pps_.ctb_ts_to_rs[ctb_ts_];
    const int cx = ctb_rs_ % W, cy = ctb_rs_ / W;
    const bool tile_first = ctb_ts_ == 0 || pps_.tile_id[ctb_ts_] != pps_.tile_id[ctb_ts_ - 1];
    const bool row_first = pps_.entropy_coding_sync && (cx == 0 || pps_.tile_id[ctb_ts_] != pps_.tile_id[pps_.ctb_rs_to_ts[ctb_rs_ - 1]]);
    // 9.3.1: which context variables the CTU starts from
    if (tile_first) {
      init_contexts();
      st_.last_qp = sh.qp;
    } else if (row_first) {
      const int x0 = cx << sps_.log2_ctb, y0 = cy << sps_.log2_ctb;
      pic_.slice_addr[ctb_rs_] = sh.slice_address;          // the availability test below looks at the current CTB too
      if (st_.wpp_valid && available(x0, y0, x0 + sps_.ctb_size(), y0 - sps_.ctb_size())) ctx_ = st_.wpp;
      else init_contexts();
      st_.last_qp = sh.qp;
    } else if (first) {
      if (sh.dependent) {
        if (!st_.have_end_of_segment) throw ParseError("dependent slice segment without stored context variables");
        ctx_ = st_.end_of_segment;
      } else {
        init_contexts();
      }
    }
    first = false;
    ctu(ctb_rs_);
    if (pps_.entropy_coding_sync) {                         // 9.3.2.4: storage after the second CTB of a row of the tile
      int col_start = 0;
      for (int v : pps_.col_bd) if (v <= cx) col_start = v;
      if (cx == col_start + 1) { st_.wpp = ctx_; st_.wpp_valid = true; }
    }
    if (!pps_.tiles_enabled && cx == W - 1) {               // a CTB row is complete: later pictures may take motion vectors from it
      if (hooks_ && hooks_->rows_done) hooks_->rows_done(hooks_->self, &pic_, cy + 1);
      else pic_.rows_done.store(cy + 1, std::memory_order_release);
    }
    const int end = cabac_.terminate();                     // end_of_slice_segment_flag
    ctb_ts_++;
    if (end) {
      cabac_.finish_to_byte();                              // rbsp_slice_segment_trailing_bits(): catches a parser that lost sync
      if (pps_.dependent_slice_segments_enabled) { st_.end_of_segment = ctx_; st_.have_end_of_segment = true; }
      done = true;
    } else {
      if (ctb_ts_ >= pic_.num_ctbs) throw ParseError("slice data runs past the last CTB of the picture");
      const int next_rs = pps_.ctb_ts_to_rs[ctb_ts_];
      const bool new_tile = pps_.tiles_enabled && pps_.tile_id[ctb_ts_] != pps_.tile_id[ctb_ts_ - 1];
      const bool new_row = pps_.entropy_coding_sync && (next_rs % W == 0 || pps_.tile_id[ctb_ts_] != pps_.tile_id[pps_.ctb_rs_to_ts[next_rs - 1]]);
      if (new_tile || new_row) {
        if (!cabac_.terminate()) throw ParseError("end_of_subset_one_bit is not 1");
        cabac_.finish_to_byte();
        cabac_.start(cabac_.bit_pos());
      }
    }
  }
  st_.next_ctb_ts = ctb_ts_;
  return ctb_ts_ >= pic_.num_ctbs;
}

void SliceDecoder::decode_wpp_row(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes, int row, WppShared& w) {
  sh_ = &sh;
  slice_idx_ = slice_idx;
  slice_ = pic_.slices[slice_idx].get();
  cabac_.attach(rbsp, bytes);
  no_backward_pred_ = true;
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < sh.num_ref_idx[l]; i++) if (slice_->ref_poc[l][i] > pic_.poc) no_backward_pred_ = false;
  const int W = w.width;
  int cx = 0;
  try {
    if (row > 0) w.wait(row - 1, std::min(2, W));
    if (w.failed.load()) throw ParseError("a row above could not be parsed");
    // 9.3.1: the first CTB of a row starts from the variables stored after the second CTB of the row above, if that CTB is
    // available (same slice, picture wider than one CTB); otherwise from the initial values
    if (row > 0 && W > 1) ctx_ = w.ctx_after2[row - 1];
    else init_contexts();
    st_.last_qp = sh.qp;                                      // first quantisation group of a CTB row (8.6.1)
    cabac_.start(w.row_bit_pos[row]);
    for (cx = 0; cx < W; cx++) {
      if (row > 0) w.wait(row - 1, std::min(cx + 2, W));      // the CTB above and to the right must be complete
      if (w.failed.load()) throw ParseError("a row above could not be parsed");
      ctb_rs_ = ctb_ts_ = row * W + cx;
      ctu(ctb_rs_);
      if (cx == 1) w.ctx_after2[row] = ctx_;
      if (cx == W - 1) {
        if (hooks_ && hooks_->rows_done) hooks_->rows_done(hooks_->self, &pic_, row + 1);
        else pic_.rows_done.store(row + 1, std::memory_order_release);
      }
      const int end = cabac_.terminate();                     // end_of_slice_segment_flag
      const bool last = row == w.rows - 1 && cx == W - 1;
      if ((end != 0) != last) throw ParseError("end_of_slice_segment_flag does not match the picture's last CTB");
      if (cx == W - 1) {
        if (!last && !cabac_.terminate()) throw ParseError("end_of_subset_one_bit is not 1");
        cabac_.finish_to_byte();
      }
      w.publish(row, cx + 1);
    }
  } catch (const std::exception& e) {
    { std::lock_guard<std::mutex> lk(w.mu); if (w.error.empty()) w.error = e.what(); }
    w.failed.store(true);
    for (int x = cx; x < W; x++) {                           // what this row did not finish goes back to "never decoded"
      const int rs = row * W + x;
      pic_.reset_ctu(rs);
      pic_.slice_addr[rs] = -1;
    }
    if (hooks_ && hooks_->rows_done) hooks_->rows_done(hooks_->self, &pic_, row + 1);
    w.publish(row, W);                                        // nobody below waits for this row any longer
  }
}

void SliceDecoder::decode_tile(const SliceHeader& sh, int slice_idx, const uint8_t* rbsp, size_t bytes, int tile, WppShared& w) {
  sh_ = &sh;
  slice_idx_ = slice_idx;
  slice_ = pic_.slices[slice_idx].get();
  cabac_.attach(rbsp, bytes);
  no_backward_pred_ = true;
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < sh.num_ref_idx[l]; i++) if (slice_->ref_poc[l][i] > pic_.poc) no_backward_pred_ = false;
  int first_ts = -1, last_ts = -1;                            // the tile's CTBs are consecutive in tile scan (6.5.1)
  for (int ts = 0; ts < pic_.num_ctbs; ts++) if (pps_.tile_id[ts] == tile) { if (first_ts < 0) first_ts = ts; last_ts = ts; }
  int ts = first_ts;
  try {
    if (first_ts < 0) throw ParseError("tile without CTBs");
    init_contexts();
    st_.last_qp = sh.qp;                                      // first quantisation group of a tile (8.6.1)
    cabac_.start(w.row_bit_pos[tile]);
    for (ts = first_ts; ts <= last_ts; ts++) {
      ctb_ts_ = ts;
      ctb_rs_ = pps_.ctb_ts_to_rs[ts];
      ctu(ctb_rs_);
      const int end = cabac_.terminate();                     // end_of_slice_segment_flag
      const bool last = ts == pic_.num_ctbs - 1;
      if ((end != 0) != last) throw ParseError("end_of_slice_segment_flag does not match the picture's last CTB");
      if (ts == last_ts) {
        if (!last && !cabac_.terminate()) throw ParseError("end_of_subset_one_bit is not 1");
        cabac_.finish_to_byte();
      }
    }
  } catch (const std::exception& e) {
    { std::lock_guard<std::mutex> lk(w.mu); if (w.error.empty()) w.error = e.what(); }
    w.failed.store(true);
    for (int t2 = std::max(ts, first_ts); t2 >= 0 && t2 <= last_ts; t2++) {
      const int rs = pps_.ctb_ts_to_rs[t2];
      pic_.reset_ctu(rs);
      pic_.slice_addr[rs] = -1;
    }
  }
}

void SliceDecoder::ctu(int rs) {
  const int x0 = (rs % pic_.ctbs_w) << sps_.log2_ctb, y0 = (rs / pic_.ctbs_w) << sps_.log2_ctb;
  pic_.reset_ctu(rs);
  pic_.slice_addr[rs] = sh_->slice_address;
  pic_.slice_idx[rs] = (uint16_t)slice_idx_;
  pic_.tile_idx[rs] = (uint16_t)pps_.tile_id[ctb_ts_];
  if (sh_->sao_luma || sh_->sao_chroma) sao_syntax(rs);
  coding_quadtree(x0, y0, sps_.log2_ctb, 0);
}

// 7.3.8.3 sao(); filled the way HM's parser leaves SAOBlkParam (TDecSbac.cpp:1708-1848): merge resolution happens on the device side
void SliceDecoder::sao_syntax(int rs) {
  const int rx = rs % pic_.ctbs_w, ry = rs / pic_.ctbs_w;
  hmgpu_sao_param* prm = &pic_.sao[(size_t)rs * 3];
  const bool enabled[3] = {sh_->sao_luma, sh_->sao_chroma, sh_->sao_chroma};
  bool merge_left = false, merge_up = false;
  if (rx > 0) {
    const bool in_tile = pps_.tile_id[ctb_ts_] == pps_.tile_id[pps_.ctb_rs_to_ts[rs - 1]];
    if (in_tile && pic_.slice_addr[rs - 1] == sh_->slice_address) merge_left = cabac_.decision(ctx_.s[CTX_SAO_MERGE]);
  }
  if (ry > 0 && !merge_left) {
    const bool in_tile = pps_.tile_id[ctb_ts_] == pps_.tile_id[pps_.ctb_rs_to_ts[rs - pic_.ctbs_w]];
    if (in_tile && pic_.slice_addr[rs - pic_.ctbs_w] == sh_->slice_address) merge_up = cabac_.decision(ctx_.s[CTX_SAO_MERGE]);
  }
  if (merge_left || merge_up) {
    for (int c = 0; c < 3; c++) {
      prm[c].mode_idc = enabled[c] ? HMGPU_SAO_MERGE : HMGPU_SAO_OFF;
      prm[c].type_idc = merge_left ? HMGPU_SAO_MERGE_LEFT : HMGPU_SAO_MERGE_ABOVE;
    }
    return;
  }
  for (int c = 0; c < 3; c++) {
    hmgpu_sao_param& p = prm[c];
    p = hmgpu_sao_param{};
    if (!enabled[c]) continue;
    if (c < 2) {
      int type = 0;                                   // sao_type_idx_*: TR cMax 2, first bin context coded
      if (cabac_.decision(ctx_.s[CTX_SAO_TYPE])) type = cabac_.bypass() ? 2 : 1;
      if (type == 0) continue;
      p.mode_idc = HMGPU_SAO_NEW;
      p.type_idc = type == 1 ? HMGPU_SAO_BO : HMGPU_SAO_EO_0;
    } else {
      p.mode_idc = prm[1].mode_idc;
      p.type_idc = prm[1].type_idc;
      if (p.mode_idc != HMGPU_SAO_NEW) continue;
    }
    const int bd = c ? sps_.bit_depth_chroma : sps_.bit_depth_luma;
    const int cmax = (1 << (std::min(bd, 10) - 5)) - 1;
    int off[4];
    for (int i = 0; i < 4; i++) {
      int v = 0;
      while (v < cmax && cabac_.bypass()) v++;
      off[i] = v;
    }
    if (p.type_idc == HMGPU_SAO_BO) {
      for (int i = 0; i < 4; i++) if (off[i] && cabac_.bypass()) off[i] = -off[i];
      p.type_aux_info = cabac_.bypass_bits(5);
      for (int i = 0; i < 4; i++) p.offset[(p.type_aux_info + i) & 31] = off[i];
    } else {
      if (c == 0) p.type_idc = HMGPU_SAO_EO_0 + cabac_.bypass_bits(2);
      else if (c == 1) p.type_idc = HMGPU_SAO_EO_0 + cabac_.bypass_bits(2);
      else p.type_idc = prm[1].type_idc;
      p.offset[0] = off[0];
      p.offset[1] = off[1];
      p.offset[2] = 0;
      p.offset[3] = -off[2];
      p.offset[4] = -off[3];
    }
  }
}

// 6.4.1 z-scan order availability of the block at (xn, yn) for the block at (xc, yc)
bool SliceDecoder::available(int xc, int yc, int xn, int yn) const {
  if (xn < 0 || yn < 0 || xn >= pic_.width || yn >= pic_.height) return false;
  const int cn = pic_.ctb_at(xn, yn), cc = pic_.ctb_at(xc, yc);
  // (the tile test comes first: tiles may be parsed by different threads, and nothing of another tile is to be looked at)
  if (cn != cc && pps_.tile_id[pps_.ctb_rs_to_ts[cn]] != pps_.tile_id[pps_.ctb_rs_to_ts[cc]]) return false;
  if (pic_.slice_addr[cn] != sh_->slice_address) return false;                 // other slice, or not decoded yet
  if (cn != cc) return pps_.ctb_rs_to_ts[cn] < pps_.ctb_rs_to_ts[cc];
  const int mask = (1 << sps_.log2_ctb) - 1, n4 = pic_.zs->n4;
  const int zn = pic_.zs->r2z[((yn & mask) >> 2) * n4 + ((xn & mask) >> 2)], zc = pic_.zs->r2z[((yc & mask) >> 2) * n4 + ((xc & mask) >> 2)];
  return zn <= zc;
}

void SliceDecoder::coding_quadtree(int x0, int y0, int log2, int depth) {
  const int size = 1 << log2;
  bool split;
  if (x0 + size <= pic_.width && y0 + size <= pic_.height && log2 > sps_.log2_min_cb) {
    int inc = 0;
    if (available(x0, y0, x0 - 1, y0) && pic_.depth[pic_.part_at(x0 - 1, y0)] > depth) inc++;
    if (available(x0, y0, x0, y0 - 1) && pic_.depth[pic_.part_at(x0, y0 - 1)] > depth) inc++;
    split = cabac_.decision(ctx_.s[CTX_SPLIT_CU + inc]);
  } else {
    split = log2 > sps_.log2_min_cb;
  }
  if (pps_.cu_qp_delta_enabled && log2 >= sps_.log2_ctb - pps_.diff_cu_qp_delta_depth) {
    is_cu_qp_delta_coded_ = false;
    cu_qp_delta_val_ = 0;
    start_quant_group(x0, y0);
  }
  if (split) {
    const int h = size >> 1;
    coding_quadtree(x0, y0, log2 - 1, depth + 1);
    if (x0 + h < pic_.width) coding_quadtree(x0 + h, y0, log2 - 1, depth + 1);
    if (y0 + h < pic_.height) coding_quadtree(x0, y0 + h, log2 - 1, depth + 1);
    if (x0 + h < pic_.width && y0 + h < pic_.height) coding_quadtree(x0 + h, y0 + h, log2 - 1, depth + 1);
  } else {
    coding_unit(x0, y0, log2, depth);
  }
}

// 8.6.1: qPY_PRED of the quantisation group starting at (x0, y0)
void SliceDecoder::start_quant_group(int x0, int y0) {
  qg_x_ = x0;
  qg_y_ = y0;
  const int mask = (1 << sps_.log2_ctb) - 1;
  const int a = (x0 & mask) ? pic_.qp[pic_.part_at(x0 - 1, y0)] : st_.last_qp;
  const int b = (y0 & mask) ? pic_.qp[pic_.part_at(x0, y0 - 1)] : st_.last_qp;
  qg_pred_ = (a + b + 1) >> 1;
}

int SliceDecoder::cu_qp() const {
  if (!pps_.cu_qp_delta_enabled) return sh_->qp;
  const int off = 6 * (sps_.bit_depth_luma - 8);
  return ((qg_pred_ + cu_qp_delta_val_ + 52 + 2 * off) % (52 + off)) - off;
}

// 7.3.8.14 cu_qp_delta_abs / cu_qp_delta_sign_flag
void SliceDecoder::qp_delta() {
  int v = 0;
  if (cabac_.decision(ctx_.s[CTX_QP_DELTA])) {
    v = 1;
    while (v < 5 && cabac_.decision(ctx_.s[CTX_QP_DELTA + 1])) v++;
    if (v == 5) {                                     // EG0 suffix
      int k = 0;
      unsigned s = 0;
      while (cabac_.bypass()) { s += 1u << k; if (++k > 16) throw ParseError("cu_qp_delta_abs too large"); }
      if (k) s += cabac_.bypass_bits(k);
      v += (int)s;
    }
  }
  if (v && cabac_.bypass()) v = -v;
  const int off = 6 * (sps_.bit_depth_luma - 8);
  if (v < -(26 + off / 2) || v > 25 + off / 2) throw ParseError("CuQpDeltaVal out of range");
  cu_qp_delta_val_ = v;
  is_cu_qp_delta_coded_ = true;
}

// ------------------------------------------------------------------------------------------------ coding unit
void SliceDecoder::coding_unit(int x0, int y0, int log2, int depth) {
  const int size = 1 << log2, nparts = 1 << (2 * (log2 - 2));
  const size_t base = pic_.part_at(x0, y0);
  cu_x_ = x0; cu_y_ = y0; cu_log2_ = log2;
  cu_bypass_ = pps_.transquant_bypass_enabled && cabac_.decision(ctx_.s[CTX_TQ_BYPASS]);
  bool skip = false;
  if (sh_->type != SLICE_I) {
    int inc = 0;
    if (available(x0, y0, x0 - 1, y0) && pic_.skip[pic_.part_at(x0 - 1, y0)]) inc++;
    if (available(x0, y0, x0, y0 - 1) && pic_.skip[pic_.part_at(x0, y0 - 1)]) inc++;
    skip = cabac_.decision(ctx_.s[CTX_SKIP + inc]);
  }
  fill_z(pic_.depth, base, nparts, (uint8_t)depth);
  fill_z(pic_.skip, base, nparts, (uint8_t)skip);
  if (cu_bypass_) { fill_z(pic_.bypass, base, nparts, (uint8_t)1); pic_.has_bypass = true; }
  int part_mode = PART_2Nx2N;
  bool pcm = false;
  if (skip) {
    cu_pred_mode_ = MODE_INTER;
    fill_z(pic_.pred_mode, base, nparts, (int8_t)MODE_INTER);
    fill_z(pic_.part_size, base, nparts, (int8_t)PART_2Nx2N);
    prediction_unit(x0, y0, size, x0, y0, size, size, 0, PART_2Nx2N, depth, true);
  } else {
    cu_pred_mode_ = MODE_INTRA;
    if (sh_->type != SLICE_I) cu_pred_mode_ = cabac_.decision(ctx_.s[CTX_PRED_MODE]) ? MODE_INTRA : MODE_INTER;
    if (cu_pred_mode_ == MODE_INTRA) {
      if (log2 == sps_.log2_min_cb) part_mode = cabac_.decision(ctx_.s[CTX_PART_MODE]) ? PART_2Nx2N : PART_NxN;
    } else {                                                               // 9.3.3.6 binarisation of part_mode
      if (cabac_.decision(ctx_.s[CTX_PART_MODE])) {
        part_mode = PART_2Nx2N;
      } else if (log2 == sps_.log2_min_cb) {
        if (log2 == 3) part_mode = cabac_.decision(ctx_.s[CTX_PART_MODE + 1]) ? PART_2NxN : PART_Nx2N;
        else if (cabac_.decision(ctx_.s[CTX_PART_MODE + 1])) part_mode = PART_2NxN;
        else part_mode = cabac_.decision(ctx_.s[CTX_PART_MODE + 2]) ? PART_Nx2N : PART_NxN;
      } else if (!sps_.amp) {
        part_mode = cabac_.decision(ctx_.s[CTX_PART_MODE + 1]) ? PART_2NxN : PART_Nx2N;
      } else {
        const bool hor = cabac_.decision(ctx_.s[CTX_PART_MODE + 1]);
        if (cabac_.decision(ctx_.s[CTX_PART_MODE + 3])) part_mode = hor ? PART_2NxN : PART_Nx2N;
        else if (cabac_.bypass()) part_mode = hor ? PART_2NxnD : PART_nRx2N;
        else part_mode = hor ? PART_2NxnU : PART_nLx2N;
      }
    }
    fill_z(pic_.pred_mode, base, nparts, (int8_t)cu_pred_mode_);
    fill_z(pic_.part_size, base, nparts, (int8_t)part_mode);
    if (cu_pred_mode_ == MODE_INTRA) {
      if (part_mode == PART_2Nx2N && sps_.pcm && log2 >= sps_.log2_min_pcm_cb && log2 <= sps_.log2_max_pcm_cb) pcm = cabac_.terminate();
      if (pcm) {
        cabac_.finish_to_byte();                   // pcm_alignment_zero_bits
        pcm_sample(x0, y0, log2);
        cabac_.start(cabac_.bit_pos());
        fill_z(pic_.ipcm, base, nparts, (uint8_t)1);
        pic_.has_pcm = true;
      } else {
        intra_modes(x0, y0, log2, part_mode == PART_NxN);
      }
    } else {
      const int h = size >> 1, q = size >> 2;
      switch (part_mode) {
        case PART_2Nx2N: prediction_unit(x0, y0, size, x0, y0, size, size, 0, part_mode, depth, false); break;
        case PART_2NxN:
          prediction_unit(x0, y0, size, x0, y0, size, h, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0, y0 + h, size, h, 1, part_mode, depth, false);
          break;
        case PART_Nx2N:
          prediction_unit(x0, y0, size, x0, y0, h, size, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0 + h, y0, h, size, 1, part_mode, depth, false);
          break;
        case PART_2NxnU:
          prediction_unit(x0, y0, size, x0, y0, size, q, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0, y0 + q, size, size - q, 1, part_mode, depth, false);
          break;
        case PART_2NxnD:
          prediction_unit(x0, y0, size, x0, y0, size, size - q, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0, y0 + size - q, size, q, 1, part_mode, depth, false);
          break;
        case PART_nLx2N:
          prediction_unit(x0, y0, size, x0, y0, q, size, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0 + q, y0, size - q, size, 1, part_mode, depth, false);
          break;
        case PART_nRx2N:
          prediction_unit(x0, y0, size, x0, y0, size - q, size, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0 + size - q, y0, q, size, 1, part_mode, depth, false);
          break;
        default:
          prediction_unit(x0, y0, size, x0, y0, h, h, 0, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0 + h, y0, h, h, 1, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0, y0 + h, h, h, 2, part_mode, depth, false);
          prediction_unit(x0, y0, size, x0 + h, y0 + h, h, h, 3, part_mode, depth, false);
          break;
      }
    }
    if (!pcm) {
      bool root_cbf = true;
      if (cu_pred_mode_ != MODE_INTRA && !(part_mode == PART_2Nx2N && pic_.merge[base])) root_cbf = cabac_.decision(ctx_.s[CTX_ROOT_CBF]);
      if (root_cbf) transform_tree(x0, y0, x0, y0, log2, 0, 0, log2, false, false);
    }
  }
  const int qp = cu_qp();
  fill_z(pic_.qp, base, nparts, (int8_t)qp);
  st_.last_qp = qp;
}

// 7.3.8.7 pcm_sample()
void SliceDecoder::pcm_sample(int x0, int y0, int log2) {
  const int size = 1 << log2;
  const size_t ctb = pic_.ctb_at(x0, y0), z = pic_.part_at(x0, y0) - ctb * pic_.parts;
  int16_t* y = &pic_.pcm[0][(ctb << (2 * sps_.log2_ctb)) + 16 * z];
  for (int i = 0; i < size * size; i++) y[i] = (int16_t)cabac_.plain_bits(sps_.pcm_bit_depth_luma);
  for (int c = 1; c < 3 && sps_.chroma_format_idc != 0; c++) {
    int16_t* d = &pic_.pcm[c][(ctb << (2 * sps_.log2_ctb - pic_.cshift)) + ((16 * z) >> pic_.cshift)];
    for (int i = 0; i < (size * size) >> pic_.cshift; i++) d[i] = (int16_t)cabac_.plain_bits(sps_.pcm_bit_depth_chroma);
  }
}

// 7.3.8.5 (intra part of coding_unit) + 8.4.2 / 8.4.3
void SliceDecoder::intra_modes(int x0, int y0, int log2, bool nxn) {
  const int n = nxn ? 4 : 1, pb = (1 << log2) >> (nxn ? 1 : 0);
  bool prev[4];
  for (int i = 0; i < n; i++) prev[i] = cabac_.decision(ctx_.s[CTX_PREV_INTRA]);
  for (int i = 0; i < n; i++) {
    int mpm_idx = 0, rem = 0;
    if (prev[i]) { if (cabac_.bypass()) mpm_idx = cabac_.bypass() ? 2 : 1; }
    else rem = cabac_.bypass_bits(5);
    const int x = x0 + (i & 1) * pb, y = y0 + (i >> 1) * pb;
    auto cand = [&](int xn, int yn, bool above) {
      if (!available(x, y, xn, yn)) return 1;
      const size_t p = pic_.part_at(xn, yn);
      if (pic_.pred_mode[p] != MODE_INTRA || pic_.ipcm[p]) return 1;
      if (above && yn < ((y >> sps_.log2_ctb) << sps_.log2_ctb)) return 1;
      return (int)pic_.intra_dir[0][p];
    };
    const int a = cand(x - 1, y, false), b = cand(x, y - 1, true);
    int c[3];
    if (a == b) {
      if (a < 2) { c[0] = 0; c[1] = 1; c[2] = 26; }
      else { c[0] = a; c[1] = 2 + ((a + 29) % 32); c[2] = 2 + ((a - 2 + 1) % 32); }
    } else {
      c[0] = a; c[1] = b;
      c[2] = (a != 0 && b != 0) ? 0 : (a != 1 && b != 1) ? 1 : 26;
    }
    int mode;
    if (prev[i]) {
      mode = c[mpm_idx];
    } else {
      if (c[0] > c[1]) std::swap(c[0], c[1]);
      if (c[0] > c[2]) std::swap(c[0], c[2]);
      if (c[1] > c[2]) std::swap(c[1], c[2]);
      mode = rem;
      for (int k = 0; k < 3; k++) if (mode >= c[k]) mode++;
    }
    intra_luma_[i] = mode;
    for (int yy = y; yy < y + pb; yy += 4) for (int xx = x; xx < x + pb; xx += 4) pic_.intra_dir[0][pic_.part_at(xx, yy)] = (uint8_t)mode;
  }
  // intra_chroma_pred_mode: one per CU, in 4:4:4 one per prediction block (7.3.8.5; HM: enable4ChromaPUsInIntraNxNCU, TComChromaFormat.h:118-121)
  const int nc = (sps_.chroma_format_idc == 3 && nxn) ? 4 : 1;
  const size_t base = pic_.part_at(x0, y0);
  const int cu_parts = 1 << (2 * (log2 - 2));
  for (int i = 0; i < nc; i++) {
    int chroma = 4;
    if (sps_.chroma_format_idc != 0 && cabac_.decision(ctx_.s[CTX_CHROMA_MODE])) chroma = cabac_.bypass_bits(2);     // (TDecEntropy.cpp:119)
    int stored = kDmChroma;
    if (chroma != 4) {
      static const int kModes[4] = {0, 26, 10, 1};
      stored = kModes[chroma] == intra_luma_[i] ? 34 : kModes[chroma];
    }
    if (nc == 1) fill_z(pic_.intra_dir[1], base, cu_parts, (uint8_t)stored);
    else fill_z(pic_.intra_dir[1], base + (size_t)i * (cu_parts >> 2), cu_parts >> 2, (uint8_t)stored);
  }
}

// the prediction mode the chroma block at partition `part` ends up with (8.4.3; HM: TDecCu.cpp:523-525, getChromasCorrespondingPULumaIdx):
// "same as luma" resolved -- 4:4:4: the luma mode of the same partition; else the one of the CU's first partition --, then the 4:2:2 table
int SliceDecoder::chroma_pred_mode(size_t part) const {
  static const uint8_t k422[35] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31};
  int mode = pic_.intra_dir[1][part];
  if (mode == kDmChroma) {
    const size_t cu_first = pic_.part_at(cu_x_, cu_y_);
    mode = pic_.intra_dir[0][sps_.chroma_format_idc == 3 ? part : cu_first];
  }
  if (sps_.chroma_format_idc == 2 && mode < 35) mode = k422[mode];
  return mode;
}

// ------------------------------------------------------------------------------------------------ prediction unit
SliceDecoder::Motion SliceDecoder::motion_at(int x, int y) const {
  const size_t p = pic_.part_at(x, y);
  Motion m;
  for (int l = 0; l < 2; l++) {
    m.ref[l] = pic_.ref_idx[l][p];
    m.mv[l].x = pic_.mv[l][2 * p];
    m.mv[l].y = pic_.mv[l][2 * p + 1];
  }
  return m;
}

void SliceDecoder::set_motion(int x0, int y0, int w, int h, const Motion& m) {
  for (int y = y0; y < y0 + h; y += 4)
    for (int x = x0; x < x0 + w; x += 4) {
      const size_t p = pic_.part_at(x, y);
      for (int l = 0; l < 2; l++) {
        pic_.ref_idx[l][p] = m.ref[l];
        pic_.mv[l][2 * p] = m.ref[l] >= 0 ? m.mv[l].x : 0;
        pic_.mv[l][2 * p + 1] = m.ref[l] >= 0 ? m.mv[l].y : 0;
      }
    }
}

// 6.4.2 availability of a neighbouring prediction block (also false for intra neighbours)
bool SliceDecoder::pu_available(int xcb, int ycb, int ncbs, int xpb, int ypb, int w, int h, int part_idx, int xn, int yn) const {
  bool ok;
  const bool same_cb = xcb <= xn && ycb <= yn && xcb + ncbs > xn && ycb + ncbs > yn;
  if (!same_cb) ok = available(xpb, ypb, xn, yn);
  else ok = !((w << 1) == ncbs && (h << 1) == ncbs && part_idx == 1 && ycb + h <= yn && xcb + w > xn);
  if (ok && pic_.pred_mode[pic_.part_at(xn, yn)] != MODE_INTER) ok = false;
  return ok;
}

SliceDecoder::Mv SliceDecoder::scale_mv(Mv mv, int tb, int td) {
  td = clip3(-128, 127, td);
  tb = clip3(-128, 127, tb);
  const int tx = (16384 + (std::abs(td) >> 1)) / td;
  const int f = clip3(-4096, 4095, (tb * tx + 32) >> 6);
  auto one = [f](int v) {
    const int p = f * v;
    return (int16_t)clip3(-32768, 32767, (p < 0 ? -1 : 1) * ((std::abs(p) + 127) >> 8));
  };
  Mv r;
  r.x = one(mv.x);
  r.y = one(mv.y);
  return r;
}

// 8.5.3.2.9: motion of the collocated block covering (xcol, ycol) (already on the 16x16 grid) for list `list`, reference `ref_idx`
bool SliceDecoder::col_mv(int xcol, int ycol, int list, int ref_idx, Mv& out) const {
  const PicData* col = slice_->ref_pics[(sh_->type == SLICE_B && !sh_->collocated_from_l0) ? 1 : 0][sh_->collocated_ref_idx];
  if (!col) return false;
  if (hooks_ && hooks_->wait_rows && col->rows_done.load(std::memory_order_acquire) <= (ycol >> sps_.log2_ctb))
    hooks_->wait_rows(hooks_->self, col, (ycol >> sps_.log2_ctb) + 1);      // the collocated picture is still being parsed
  const size_t p = col->part_at(xcol, ycol);
  if (col->pred_mode[p] != MODE_INTER) return false;
  int lc;
  if (col->ref_idx[0][p] < 0) lc = 1;
  else if (col->ref_idx[1][p] < 0) lc = 0;
  else lc = no_backward_pred_ ? list : (sh_->collocated_from_l0 ? 1 : 0);
  const int ref_col = col->ref_idx[lc][p];
  if (ref_col < 0) return false;
  const size_t cslice = col->slice_idx[col->ctb_at(xcol, ycol)];
  if (cslice >= col->slices.size()) return false;          // (a damaged collocated picture)
  const SliceInfo& cs = *col->slices[cslice];
  const bool cur_lt = slice_->ref_is_lt[list][ref_idx], col_lt = cs.ref_is_lt[lc][ref_col];
  if (cur_lt != col_lt) return false;
  Mv mv;
  mv.x = col->mv[lc][2 * p];
  mv.y = col->mv[lc][2 * p + 1];
  const int col_diff = col->poc - cs.ref_poc[lc][ref_col], cur_diff = pic_.poc - slice_->ref_poc[list][ref_idx];
  if (cur_lt || col_diff == cur_diff || col_diff == 0) out = mv;
  else out = scale_mv(mv, cur_diff, col_diff);
  return true;
}

// 8.5.3.2.8 temporal luma motion vector prediction
bool SliceDecoder::temporal_mv(int xpb, int ypb, int w, int h, int list, int ref_idx, Mv& out) const {
  if (!sh_->temporal_mvp) return false;
  const int xbr = xpb + w, ybr = ypb + h;
  if ((ypb >> sps_.log2_ctb) == (ybr >> sps_.log2_ctb) && ybr < pic_.height && xbr < pic_.width)
    if (col_mv((xbr >> 4) << 4, (ybr >> 4) << 4, list, ref_idx, out)) return true;
  const int xc = xpb + (w >> 1), yc = ypb + (h >> 1);
  return col_mv((xc >> 4) << 4, (yc >> 4) << 4, list, ref_idx, out);
}

// 8.5.3.2.2 - 8.5.3.2.5 merge mode
void SliceDecoder::merge_candidates(int xcb, int ycb, int ncbs, int xpb, int ypb, int w, int h, int part_idx, int part_mode, int merge_idx, Motion& out) {
  const int ow = w, oh = h;
  if (pps_.log2_par_mrg_level > 2 && ncbs == 8) { xpb = xcb; ypb = ycb; w = h = ncbs; part_idx = 0; part_mode = PART_2Nx2N; }
  const int L = pps_.log2_par_mrg_level, maxc = sh_->max_num_merge_cand;
  Motion cand[6];
  int n = 0;
  auto usable = [&](int xn, int yn) {
    if ((xpb >> L) == (xn >> L) && (ypb >> L) == (yn >> L)) return false;
    return pu_available(xcb, ycb, ncbs, xpb, ypb, w, h, part_idx, xn, yn);
  };
  bool av_a1 = false, av_b1 = false;
  Motion a1, b1;
  int spatial = 0;
  if (!((part_mode == PART_Nx2N || part_mode == PART_nLx2N || part_mode == PART_nRx2N) && part_idx == 1) && usable(xpb - 1, ypb + h - 1)) {
    a1 = motion_at(xpb - 1, ypb + h - 1);
    av_a1 = true;
    cand[n++] = a1;
    spatial++;
  }
  if (!((part_mode == PART_2NxN || part_mode == PART_2NxnU || part_mode == PART_2NxnD) && part_idx == 1) && usable(xpb + w - 1, ypb - 1)) {
    b1 = motion_at(xpb + w - 1, ypb - 1);
    av_b1 = true;                            // stays "available" for the comparisons below even when it duplicates A1 (HM: isAvailableB1)
    if (!(av_a1 && b1.same(a1))) { cand[n++] = b1; spatial++; }
  }
  if (usable(xpb + w, ypb - 1)) {
    const Motion b0 = motion_at(xpb + w, ypb - 1);
    if (!(av_b1 && b0.same(b1))) { cand[n++] = b0; spatial++; }
  }
  if (usable(xpb - 1, ypb + h)) {
    const Motion a0 = motion_at(xpb - 1, ypb + h);
    if (!(av_a1 && a0.same(a1))) { cand[n++] = a0; spatial++; }
  }
  if (spatial != 4 && usable(xpb - 1, ypb - 1)) {
    const Motion b2 = motion_at(xpb - 1, ypb - 1);
    if (!(av_a1 && b2.same(a1)) && !(av_b1 && b2.same(b1))) cand[n++] = b2;
  }
  if (n < maxc && sh_->temporal_mvp) {
    Motion c;
    Mv mv;
    bool any = false;
    if (temporal_mv(xpb, ypb, w, h, 0, 0, mv)) { c.mv[0] = mv; c.ref[0] = 0; any = true; }
    if (sh_->type == SLICE_B && temporal_mv(xpb, ypb, w, h, 1, 0, mv)) { c.mv[1] = mv; c.ref[1] = 0; any = true; }
    if (any) cand[n++] = c;
  }
  if (n > maxc) n = maxc;
  if (sh_->type == SLICE_B && n > 1 && n < maxc) {                      // 8.5.3.2.4 combined bi-predictive candidates
    static const uint8_t l0c[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1c[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
    const int orig = n;
    for (int k = 0; k < orig * (orig - 1) && n < maxc; k++) {
      const Motion &p0 = cand[l0c[k]], &p1 = cand[l1c[k]];
      if (p0.ref[0] >= 0 && p1.ref[1] >= 0 && (slice_->ref_poc[0][p0.ref[0]] != slice_->ref_poc[1][p1.ref[1]] || !(p0.mv[0] == p1.mv[1]))) {
        Motion c;
        c.mv[0] = p0.mv[0]; c.ref[0] = p0.ref[0];
        c.mv[1] = p1.mv[1]; c.ref[1] = p1.ref[1];
        cand[n++] = c;
      }
    }
  }
  const int num_ref = sh_->type == SLICE_P ? sh_->num_ref_idx[0] : std::min(sh_->num_ref_idx[0], sh_->num_ref_idx[1]);
  for (int zero = 0; n < maxc; zero++) {                                // 8.5.3.2.5 zero candidates
    Motion c;
    c.ref[0] = (int8_t)(zero < num_ref ? zero : 0);
    if (sh_->type == SLICE_B) c.ref[1] = c.ref[0];
    cand[n++] = c;
  }
  out = cand[merge_idx];
  if (out.ref[0] >= 0 && out.ref[1] >= 0 && ow + oh == 12) { out.ref[1] = -1; out.mv[1] = Mv(); }
}

// 8.5.3.2.6 / 8.5.3.2.7 luma motion vector prediction
SliceDecoder::Mv SliceDecoder::amvp(int xcb, int ycb, int ncbs, int xpb, int ypb, int w, int h, int part_idx, int list, int ref_idx, int mvp_flag) {
  const int X = list, Y = 1 - list;
  const int target_poc = slice_->ref_poc[X][ref_idx];
  const bool target_lt = slice_->ref_is_lt[X][ref_idx];
  auto same_poc = [&](const Motion& m, Mv& mv) {          // first pass: a motion vector pointing at the very same picture
    if (m.ref[X] >= 0 && slice_->ref_poc[X][m.ref[X]] == target_poc) { mv = m.mv[X]; return true; }
    if (m.ref[Y] >= 0 && slice_->ref_poc[Y][m.ref[Y]] == target_poc) { mv = m.mv[Y]; return true; }
    return false;
  };
  auto scaled = [&](const Motion& m, Mv& mv) {            // second pass: any motion vector of the same kind of reference, scaled
    for (int l : {X, Y}) {
      if (m.ref[l] < 0) continue;
      const bool lt = slice_->ref_is_lt[l][m.ref[l]];
      if (lt != target_lt) continue;
      mv = m.mv[l];
      const int nb_poc = slice_->ref_poc[l][m.ref[l]];
      if (!lt && nb_poc != target_poc) mv = scale_mv(mv, pic_.poc - target_poc, pic_.poc - nb_poc);
      return true;
    }
    return false;
  };
  const int ax[2] = {xpb - 1, xpb - 1}, ay[2] = {ypb + h, ypb + h - 1};
  const int bx[3] = {xpb + w, xpb + w - 1, xpb - 1}, by[3] = {ypb - 1, ypb - 1, ypb - 1};
  bool av_a[2], av_b[3];
  for (int k = 0; k < 2; k++) av_a[k] = pu_available(xcb, ycb, ncbs, xpb, ypb, w, h, part_idx, ax[k], ay[k]);
  for (int k = 0; k < 3; k++) av_b[k] = pu_available(xcb, ycb, ncbs, xpb, ypb, w, h, part_idx, bx[k], by[k]);
  const bool is_scaled = av_a[0] || av_a[1];
  bool have_a = false, have_b = false;
  Mv mva, mvb;
  for (int k = 0; k < 2 && !have_a; k++) if (av_a[k]) have_a = same_poc(motion_at(ax[k], ay[k]), mva);
  for (int k = 0; k < 2 && !have_a; k++) if (av_a[k]) have_a = scaled(motion_at(ax[k], ay[k]), mva);
  for (int k = 0; k < 3 && !have_b; k++) if (av_b[k]) have_b = same_poc(motion_at(bx[k], by[k]), mvb);
  if (!is_scaled) {
    if (have_b) { mva = mvb; have_a = true; }
    have_b = false;
    for (int k = 0; k < 3 && !have_b; k++) if (av_b[k]) have_b = scaled(motion_at(bx[k], by[k]), mvb);
  }
  Mv list2[3];
  int n = 0;
  if (have_a) list2[n++] = mva;
  if (have_b && !(have_a && mva == mvb)) list2[n++] = mvb;
  if (n < 2) {
    Mv col;
    if (temporal_mv(xpb, ypb, w, h, X, ref_idx, col)) list2[n++] = col;
  }
  while (n < 2) list2[n++] = Mv();
  return list2[mvp_flag];
}

void SliceDecoder::prediction_unit(int xcb, int ycb, int ncbs, int x0, int y0, int w, int h, int part_idx, int part_mode, int depth, bool skip) {
  const bool merge = skip || cabac_.decision(ctx_.s[CTX_MERGE_FLAG]);
  Motion m;
  int merge_idx = 0, inter_dir = 0;
  if (merge) {
    if (sh_->max_num_merge_cand > 1 && cabac_.decision(ctx_.s[CTX_MERGE_IDX])) {
      merge_idx = 1;
      while (merge_idx < sh_->max_num_merge_cand - 1 && cabac_.bypass()) merge_idx++;
    }
    merge_candidates(xcb, ycb, ncbs, x0, y0, w, h, part_idx, part_mode, merge_idx, m);
  } else {
    int dir = PRED_L0;
    if (sh_->type == SLICE_B) {
      if (w + h != 12 && cabac_.decision(ctx_.s[CTX_INTER_DIR + depth])) dir = PRED_BI;
      else dir = cabac_.decision(ctx_.s[CTX_INTER_DIR + 4]) ? PRED_L1 : PRED_L0;
    }
    auto mvd_coding = [&](int& dx, int& dy) {                           // 7.3.8.9
      const bool g0x = cabac_.decision(ctx_.s[CTX_MVD_GT0]), g0y = cabac_.decision(ctx_.s[CTX_MVD_GT0]);
      const bool g1x = g0x && cabac_.decision(ctx_.s[CTX_MVD_GT1]), g1y = g0y && cabac_.decision(ctx_.s[CTX_MVD_GT1]);
      auto rest = [&](bool g0, bool g1) {
        if (!g0) return 0;
        int a = 1;
        if (g1) {                                                       // abs_mvd_minus2: EG1
          int k = 1;
          unsigned s = 0;
          while (cabac_.bypass()) { s += 1u << k; if (++k > 16) throw ParseError("abs_mvd_minus2 too large"); }
          s += cabac_.bypass_bits(k);
          a = 2 + (int)s;
        }
        return cabac_.bypass() ? -a : a;
      };
      dx = rest(g0x, g1x);
      dy = rest(g0y, g1y);
    };
    for (int l = 0; l < 2; l++) {
      if (dir == (l == 0 ? PRED_L1 : PRED_L0)) continue;
      int ref = 0;
      const int cmax = sh_->num_ref_idx[l] - 1;
      while (ref < cmax) {
        const int bit = ref < 2 ? cabac_.decision(ctx_.s[CTX_REF_IDX + ref]) : cabac_.bypass();
        if (!bit) break;
        ref++;
      }
      int dx = 0, dy = 0;
      if (!(l == 1 && sh_->mvd_l1_zero && dir == PRED_BI)) mvd_coding(dx, dy);
      const int mvp_flag = cabac_.decision(ctx_.s[CTX_MVP]);
      const Mv p = amvp(xcb, ycb, ncbs, x0, y0, w, h, part_idx, l, ref, mvp_flag);
      m.ref[l] = (int8_t)ref;
      m.mv[l].x = (int16_t)(p.x + dx);
      m.mv[l].y = (int16_t)(p.y + dy);
    }
  }
  inter_dir = (m.ref[0] >= 0 ? 1 : 0) | (m.ref[1] >= 0 ? 2 : 0);
  set_motion(x0, y0, w, h, m);
  for (int y = y0; y < y0 + h; y += 4)
    for (int x = x0; x < x0 + w; x += 4) {
      const size_t p = pic_.part_at(x, y);
      pic_.merge[p] = merge;
      pic_.merge_idx[p] = (uint8_t)merge_idx;
      pic_.inter_dir[p] = (uint8_t)inter_dir;
    }
}

// ------------------------------------------------------------------------------------------------ transform tree
// parent_cb / parent_cr: the chroma flags of the node above (4:2:2: bit 0 the upper, bit 1 the lower of its two squares)
void SliceDecoder::transform_tree(int x0, int y0, int xbase, int ybase, int log2, int tr_depth, int blk, int cu_log2, int parent_cb, int parent_cr) {
  const size_t base = pic_.part_at(x0, y0);
  const int nparts = 1 << (2 * (log2 - 2));
  const int8_t part_mode = pic_.part_size[base];
  const int fmt = sps_.chroma_format_idc;
  const bool intra_split = cu_pred_mode_ == MODE_INTRA && part_mode == PART_NxN;
  const int max_depth = cu_pred_mode_ == MODE_INTRA ? sps_.max_th_depth_intra + (intra_split ? 1 : 0) : sps_.max_th_depth_inter;
  bool split;
  if (log2 <= sps_.log2_max_tb && log2 > sps_.log2_min_tb && tr_depth < max_depth && !(intra_split && tr_depth == 0)) {
    split = cabac_.decision(ctx_.s[CTX_SPLIT_TU + 5 - log2]);
  } else {
    const bool inter_split = sps_.max_th_depth_inter == 0 && cu_pred_mode_ == MODE_INTER && part_mode != PART_2Nx2N && tr_depth == 0;
    split = log2 > sps_.log2_max_tb || (intra_split && tr_depth == 0) || inter_split;
  }
  int cbf_c[2] = {parent_cb, parent_cr};                // blocks that are not divided any further: the chroma flags of the parent stand
  if (fmt == 0) cbf_c[0] = cbf_c[1] = 0;                // monochrome: no chroma blocks at all (TDecEntropy.cpp:380)
  else if (log2 > 2 || fmt == 3) {
    // 7.3.8.8: one flag per component, in 4:2:2 a second one for the lower square where the chroma block is not divided further
    const bool two = fmt == 2 && (!split || log2 == 3);
    for (int k = 0; k < 2; k++) {
      const int parent = cbf_c[k];
      cbf_c[k] = 0;
      if (tr_depth == 0 || parent) {
        cbf_c[k] = cabac_.decision(ctx_.s[CTX_CBF_CHROMA + tr_depth]) ? 1 : 0;
        if (two) cbf_c[k] |= cabac_.decision(ctx_.s[CTX_CBF_CHROMA + tr_depth]) ? 2 : 0;
      }
      // HM's layout (parseQtCbf, TDecSbac.cpp:1027-1126): the node's bit where any square is coded, in 4:2:2 one depth further down the
      // flag of the upper square over the first half of the partitions, of the lower one over the second
      if (cbf_c[k]) for (int i = 0; i < nparts; i++) pic_.cbf[1 + k][base + i] |= (uint8_t)(1 << tr_depth);
      if (two && !split) {                              // (log2 == 3 with four 4x4 luma blocks below: the blocks set the two bits, see there)
        for (int i = 0; i < nparts; i++)
          if ((cbf_c[k] >> (i >= nparts / 2 ? 1 : 0)) & 1) pic_.cbf[1 + k][base + i] |= (uint8_t)(1 << (tr_depth + 1));
      }
    }
  } else {
    // a 4x4 luma block of four: the chroma block of the 8x8 node above is coded with the last of them; the node's flags at this depth
    // too (lowestTUDepth, :1055), in 4:2:2 the squares' one further down: blocks 0, 1 lie over the upper square, 2, 3 over the lower
    for (int k = 0; k < 2; k++) {
      if (cbf_c[k]) pic_.cbf[1 + k][base] |= (uint8_t)(1 << tr_depth);
      if (fmt == 2 && ((cbf_c[k] >> (blk >> 1)) & 1)) pic_.cbf[1 + k][base] |= (uint8_t)(1 << (tr_depth + 1));
    }
  }
  if (split) {
    const int h = 1 << (log2 - 1);
    transform_tree(x0, y0, x0, y0, log2 - 1, tr_depth + 1, 0, cu_log2, cbf_c[0], cbf_c[1]);
    transform_tree(x0 + h, y0, x0, y0, log2 - 1, tr_depth + 1, 1, cu_log2, cbf_c[0], cbf_c[1]);
    transform_tree(x0, y0 + h, x0, y0, log2 - 1, tr_depth + 1, 2, cu_log2, cbf_c[0], cbf_c[1]);
    transform_tree(x0 + h, y0 + h, x0, y0, log2 - 1, tr_depth + 1, 3, cu_log2, cbf_c[0], cbf_c[1]);
    uint8_t any = 0;
    for (int i = 0; i < nparts; i++) any |= pic_.cbf[0][base + i];
    if ((any >> (tr_depth + 1)) & 1) for (int i = 0; i < nparts; i++) pic_.cbf[0][base + i] |= (uint8_t)(1 << tr_depth);
    return;
  }
  bool cbf_luma = true;
  if (cu_pred_mode_ == MODE_INTRA || tr_depth != 0 || cbf_c[0] || cbf_c[1]) cbf_luma = cabac_.decision(ctx_.s[CTX_CBF_LUMA + (tr_depth == 0 ? 1 : 0)]);
  fill_z(pic_.tr_idx, base, nparts, (uint8_t)tr_depth);
  if (cbf_luma) for (int i = 0; i < nparts; i++) pic_.cbf[0][base + i] |= (uint8_t)(1 << tr_depth);
  if (cbf_luma || cbf_c[0] || cbf_c[1]) {                   // 7.3.8.10 transform_unit()
    if (pps_.cu_qp_delta_enabled && !is_cu_qp_delta_coded_) qp_delta();
    if (cbf_luma) residual_coding(x0, y0, log2, 0);
    if (log2 > 2 || fmt == 3) {
      const int log2c = fmt == 3 ? log2 : log2 - 1;
      for (int k = 0; k < 2; k++) {
        // 7.3.8.12 cross_comp_pred(): the weight of the luma residual in this block's chroma residual (4:4:4)
        if (pps_.cross_component_prediction && fmt == 3 && cbf_luma && (cu_pred_mode_ == MODE_INTER || pic_.intra_dir[1][base] == kDmChroma)) {
          const int c0 = CTX_CCP + 5 * k;
          int alpha = 0;
          if (cabac_.decision(ctx_.s[c0])) {
            int v = 0;
            if (cabac_.decision(ctx_.s[c0 + 1])) { v = 1; if (cabac_.decision(ctx_.s[c0 + 2])) { v = 2; if (cabac_.decision(ctx_.s[c0 + 3])) v = 3; } }
            alpha = cabac_.decision(ctx_.s[c0 + 4]) ? -(1 << v) : (1 << v);
          }
          if (alpha) fill_z(pic_.ccp[k], base, nparts, (int8_t)alpha);
        }
        if (cbf_c[k] & 1) residual_coding(x0, y0, log2c, 1 + k);
        if (cbf_c[k] & 2) residual_coding(x0, y0 + (1 << log2c), log2c, 1 + k, 1);
      }
    } else if (blk == 3) {
      for (int k = 0; k < 2; k++) {
        if (cbf_c[k] & 1) residual_coding(xbase, ybase, 2, 1 + k);
        if (cbf_c[k] & 2) residual_coding(xbase, ybase + 4, 2, 1 + k, 1);
      }
    }
  }
}

// 7.3.8.11 residual_coding(); (x0, y0) in luma samples, log2 = size of the block in samples of component c
// sub: the lower square of a 4:2:2 chroma block ((x0, y0) is its own position; its levels follow the upper square's)
void SliceDecoder::residual_coding(int x0, int y0, int log2, int c, int sub) {
  Cabac eng = cabac_;                    // the engine's registers stay in CPU registers for the whole block (written back at the end)
  const int size = 1 << log2;
  const size_t part = pic_.part_at(x0, y0), ctb = pic_.ctb_at(x0, y0);
  const size_t z = (sub ? pic_.part_at(x0, y0 - size) : part) - ctb * pic_.parts;
  int16_t* dst = pic_.level_dst(c, ctb, z, size, sub);
  const bool ts = pps_.transform_skip_enabled && !cu_bypass_ && log2 <= pps_.log2_max_ts_size && eng.decision(ctx_.s[CTX_TS_FLAG + (c ? 1 : 0)]);
  const bool untransformed = ts || cu_bypass_;
  // explicit_rdpcm_flag / explicit_rdpcm_dir_flag: inter blocks that skip the transform (HM 16.0: TDecSbac.cpp:1322-1350, 1884-1917)
  int rdpcm = 0;
  if (untransformed && sps_.rext_explicit_rdpcm && cu_pred_mode_ != MODE_INTRA && eng.decision(ctx_.s[CTX_RDPCM_FLAG + (c ? 1 : 0)]))
    rdpcm = eng.decision(ctx_.s[CTX_RDPCM_DIR + (c ? 1 : 0)]) ? 2 : 1;
  if (ts || rdpcm) {
    // in 4x4 luma partitions (a 4x4 chroma block of a 4:2:0 picture lies over 2x2 of them)
    const int span_x = (size << (c ? pic_.csx : 0)) >> 2, span_y = (size << (c ? pic_.csy : 0)) >> 2;
    const uint8_t v = (uint8_t)((ts ? 1 : 0) | (rdpcm << 1));
    for (int y = 0; y < span_y; y++) for (int x = 0; x < span_x; x++) pic_.ts[c][pic_.part_at(x0 + 4 * x, y0 + 4 * y)] = v;
  }
  // last significant coefficient position (9.3.4.2.3)
  int ctx_off, ctx_shift;
  if (c == 0) { ctx_off = 3 * (log2 - 2) + ((log2 - 1) >> 2); ctx_shift = (log2 + 1) >> 2; }
  else { ctx_off = 15; ctx_shift = log2 - 2; }
  const int cmax = (log2 << 1) - 1;
  int px = 0, py = 0;
  while (px < cmax && eng.decision(ctx_.s[CTX_LAST_X + ctx_off + (px >> ctx_shift)])) px++;
  while (py < cmax && eng.decision(ctx_.s[CTX_LAST_Y + ctx_off + (py >> ctx_shift)])) py++;
  int lx = px, ly = py;
  if (px > 3) { const int nb = (px >> 1) - 1; lx = (1 << nb) * (2 + (px & 1)) + (int)eng.bypass_bits(nb); }
  if (py > 3) { const int nb = (py >> 1) - 1; ly = (1 << nb) * (2 + (py & 1)) + (int)eng.bypass_bits(nb); }
  int scan_idx = 0;
  // 7.4.9.11 scanIdx: mode-dependent scans for 4x4 blocks and for 8x8 blocks of components that are not subsampled (HM: getCoefScanIdx,
  // TComDataCU.cpp:3525-3580)
  if (cu_pred_mode_ == MODE_INTRA && (log2 == 2 || (log2 == 3 && (c == 0 || sps_.chroma_format_idc == 3)))) {
    const int mode = c == 0 ? pic_.intra_dir[0][part] : chroma_pred_mode(part);
    if (mode >= 6 && mode <= 14) scan_idx = 2;
    else if (mode >= 22 && mode <= 30) scan_idx = 1;
  }
  if (scan_idx == 2) std::swap(lx, ly);
  if (lx >= size || ly >= size) throw ParseError("last significant coefficient outside the transform block");
  const int sb_log2 = log2 - 2, sbw = 1 << sb_log2;
  const uint8_t* sb_order = kScan.order[sb_log2][scan_idx];
  const uint8_t* in_order = kScan.order[2][scan_idx];
  const int last_sb = kScan.index[sb_log2][scan_idx][(ly >> 2) * sbw + (lx >> 2)];
  const int last_pos = kScan.index[2][scan_idx][(ly & 3) * 4 + (lx & 3)];
  int csbf[9][9] = {{0}};
  int prev_c1 = 1;
  bool first_sb = true;
  // sign data hiding is off wherever RDPCM runs: explicit modes, and intra transform-skip blocks predicted along 10 / 26
  // (TDecSbac.cpp:1345-1368); a lossless CU never hides signs
  bool sdh = pps_.sign_data_hiding && !cu_bypass_ && !rdpcm;
  if (sdh && ts && sps_.rext_implicit_rdpcm && cu_pred_mode_ == MODE_INTRA) {
    const int mode = c == 0 ? pic_.intra_dir[0][part] : chroma_pred_mode(part);
    if (mode == 10 || mode == 26) sdh = false;
  }
  const bool single_sig = sps_.rext_ts_context && untransformed;      // transform_skip_context_enabled_flag (TComChromaFormat.cpp:116-120)
  uint8_t& stat = ctx_.stat_coeff[(c ? 2 : 0) + (untransformed ? 1 : 0)];   // TComTU::getGolombRiceStatisticsIndex (TComTU.cpp:236-254)
  const bool persistent_rice = sps_.rext_persistent_rice;
  for (int i = last_sb; i >= 0; i--) {
    const int xs = sb_order[i] % sbw, ys = sb_order[i] / sbw;
    const int right = csbf[ys][xs + 1], below = csbf[ys + 1][xs];
    bool coded = true, infer_dc = false;
    if (i < last_sb && i > 0) {
      coded = eng.decision(ctx_.s[CTX_CSBF + ((right | below) ? 1 : 0) + (c ? 2 : 0)]);
      infer_dc = true;
    }
    csbf[ys][xs] = coded;
    if (!coded) continue;
    // significance map (9.3.4.2.5)
    int pos[17], nsig = 0;
    int n = 15;
    if (i == last_sb) { pos[nsig++] = last_pos; n = last_pos - 1; }
    const int prev_csbf = right | (below << 1);
    // context offset shared by all positions of the sub-block (position (0,0) of the block is the exception: context 0)
    int sig_base;
    const uint8_t* sig_tab;
    if (log2 == 2) { sig_tab = kSigCtx4x4; sig_base = c ? 27 : 0; }
    else {
      sig_tab = kSigCtxPattern[prev_csbf];
      if (c == 0) sig_base = (i > 0 ? 3 : 0) + (log2 == 3 ? (scan_idx == 0 ? 9 : 15) : 21);
      else sig_base = 27 + (log2 == 3 ? 9 : 12);
    }
    for (; n >= 0; n--) {
      bool sig;
      if (n > 0 || !infer_dc) {
        const int sc = single_sig ? 42 + (c ? 1 : 0) : (n == 0 && i == 0 && log2 > 2) ? (c ? 27 : 0) : sig_base + sig_tab[in_order[n]];
        sig = eng.decision(ctx_.s[CTX_SIG + sc]);
        infer_dc = infer_dc && !sig;
      } else {
        sig = true;                                      // the only coefficient a coded sub-block can still have
      }
      pos[nsig] = n;                                       // (kept only when significant: no branch on the bin)
      nsig += sig ? 1 : 0;
    }
    if (nsig == 0) continue;
    // greater-than-1 / greater-than-2 flags (9.3.4.2.6, 9.3.4.2.7)
    int ctx_set = (i == 0 || c > 0) ? 0 : 2;
    if (!first_sb && prev_c1 == 0) ctx_set++;
    first_sb = false;
    int c1 = 1, first_g1 = -1;
    int g1[16] = {0};
    const int ng1 = nsig < 8 ? nsig : 8;
    for (int k = 0; k < ng1; k++) {
      g1[k] = eng.decision(ctx_.s[CTX_GT1 + (c ? 16 : 0) + ctx_set * 4 + c1]);
      if (g1[k]) { c1 = 0; if (first_g1 < 0) first_g1 = k; }
      else if (c1 > 0 && c1 < 3) c1++;
    }
    prev_c1 = c1;
    int g2 = 0;
    if (first_g1 >= 0) g2 = eng.decision(ctx_.s[CTX_GT2 + (c ? 4 : 0) + ctx_set]);
    const bool hidden = sdh && pos[0] - pos[nsig - 1] > 3;
    const int nsign = nsig - (hidden ? 1 : 0);
    const unsigned signs = eng.bypass_bits(nsign) << (16 - nsign);
    int rice = persistent_rice ? stat >> 2 : 0, sum = 0;
    bool first_remaining = persistent_rice;
    for (int k = 0; k < nsig; k++) {
      int level = 1 + g1[k] + (k == first_g1 ? g2 : 0);
      const int thresh = k < 8 ? (k == first_g1 ? 3 : 2) : 1;
      if (level == thresh) {
        // coeff_abs_level_remaining (9.3.3.11): unary prefix, then rice / escape suffix, all bypass bins: read as one group of 16 bins
        // (enough for prefixes up to 7 at any rice parameter), the engine keeps what the element used
        unsigned wide = 0;
        const unsigned q = rice > 4 ? 0u : eng.bypass_peek16(wide);
        const unsigned zeros = ~q & 0xffffu;
        int prefix = zeros ? __builtin_clz(zeros) - 16 : 16;
        int rem;
        if (rice > 4) {                                      // (only with persistent Rice adaptation) bin by bin
          prefix = 0;
          while (prefix < 32 && eng.bypass()) prefix++;
          if (prefix == 32) throw ParseError("coeff_abs_level_remaining prefix too long");
          if (prefix <= 3) rem = (prefix << rice) + (int)eng.bypass_bits(rice);
          else {
            if (prefix - 3 + rice > 24) throw ParseError("coeff_abs_level_remaining out of range");
            rem = (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)eng.bypass_bits(prefix - 3 + rice);
          }
        } else if (prefix <= 3) {
          const int nb = prefix + 1 + rice;
          rem = (prefix << rice) + (int)((q >> (16 - nb)) & ((1u << rice) - 1u));
          eng.bypass_keep(wide, q, nb);
        } else if (prefix <= 7) {
          const int sl = prefix - 3 + rice, nb = prefix + 1 + sl;
          rem = (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)((q >> (16 - nb)) & ((1u << sl) - 1u));
          eng.bypass_keep(wide, q, nb);
        } else {
          eng.bypass_keep(wide, q, 8);                       // eight 1 bins so far
          prefix = 8;
          while (prefix < 32 && eng.bypass()) prefix++;
          if (prefix == 32) throw ParseError("coeff_abs_level_remaining prefix too long");
          rem = (((1 << (prefix - 3)) + 3 - 1) << rice) + (int)eng.bypass_bits(prefix - 3 + rice);
        }
        level += rem;
        // (HM 16.0 leaves the parameter uncapped under persistent adaptation: TDecSbac.cpp:1576-1579)
        if (level > 3 * (1 << rice)) rice = persistent_rice ? rice + 1 : std::min(rice + 1, 4);
        if (first_remaining) {                               // StatCoeff update on the first escape of the sub-block (:1581-1595)
          const int init = stat >> 2;
          if (rem >= (3 << init)) stat++;
          else if (2 * rem < (1 << init) && stat > 0) stat--;
          first_remaining = false;
        }
      }
      sum += level;
      bool neg;
      if (k < nsign) neg = (signs >> (15 - k)) & 1;
      else neg = sum & 1;
      const int v = neg ? -level : level;
      const int xc = (xs << 2) + (in_order[pos[k]] & 3), yc = (ys << 2) + (in_order[pos[k]] >> 2);
      dst[yc * size + xc] = (int16_t)clip3(-32768, 32767, v);
    }
  }
  cabac_ = eng;
}

}  // namespace hmdec
