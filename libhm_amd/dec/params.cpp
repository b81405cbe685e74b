// params.cpp -- see params.h.  Syntax order follows Rec. ITU-T H.265 7.3.2 / 7.3.6; every function names the clause it implements.
#include "params.h"

#include <algorithm>
#include <cstring>

namespace hmdec {

static int ceil_log2(unsigned v) {
  int n = 0;
  while ((1u << n) < v) n++;
  return n;
}

// 6.5.3 up-right diagonal scan of a blk x blk array: out[i] = y * blk + x of the i-th position
static void diag_scan(int blk, int* out) {
  int i = 0, x = 0, y = 0;
  for (;;) {
    while (y >= 0) {
      if (x < blk && y < blk) out[i++] = y * blk + x;
      y--;
      x++;
    }
    y = x;
    x = 0;
    if (i >= blk * blk) break;
  }
}

// Table 7-6, in coefficient (diagonal-scan) order
static const uint8_t kDefault8x8Intra[64] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 16, 17, 16, 17, 18, 17, 18, 18, 17, 18, 21,
                                             19, 20, 21, 20, 19, 21, 24, 22, 22, 24, 24, 22, 22, 24, 25, 25, 27, 30, 27, 25, 25, 29,
                                             31, 35, 35, 31, 29, 36, 41, 44, 41, 36, 47, 54, 54, 47, 65, 70, 65, 88, 88, 115};
static const uint8_t kDefault8x8Inter[64] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 20,
                                             20, 20, 20, 20, 20, 20, 24, 24, 24, 24, 24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 28,
                                             28, 28, 28, 28, 28, 33, 33, 33, 33, 33, 41, 41, 41, 41, 54, 54, 54, 71, 71, 91};

static void default_list(int size_id, int matrix_id, int32_t* coef, int32_t* dc) {
  if (size_id == 0) {
    for (int i = 0; i < 16; i++) coef[i] = 16;
  } else {
    int scan[64];
    diag_scan(8, scan);
    const uint8_t* src = matrix_id < 3 ? kDefault8x8Intra : kDefault8x8Inter;
    for (int i = 0; i < 64; i++) coef[scan[i]] = src[i];
  }
  *dc = 16;
}

void ScalingListSet::set_default() {
  memset(coef, 0, sizeof(coef));
  for (int s = 0; s < 4; s++)
    for (int m = 0; m < 6; m++) default_list(s, m, coef[s][m], &dc[s][m]);
}

// 7.3.4 scaling_list_data()
static void parse_scaling_list_data(BitReader& br, ScalingListSet& sl) {
  int scan4[16], scan8[64];
  diag_scan(4, scan4);
  diag_scan(8, scan8);
  for (int size_id = 0; size_id < 4; size_id++) {
    const int step = size_id == 3 ? 3 : 1;
    for (int m = 0; m < 6; m++) {
      if (size_id == 3 && m % 3) {         // 4:2:0 never uses them; kept as HM keeps them (copy of the 16x16 list)
        memcpy(sl.coef[3][m], sl.coef[2][m], sizeof(sl.coef[3][m]));
        sl.dc[3][m] = sl.dc[2][m];
        continue;
      }
      const int n = size_id == 0 ? 16 : 64;
      if (!br.flag()) {                    // scaling_list_pred_mode_flag == 0: copy a reference list or the default
        const unsigned delta = br.ue();
        if (delta * step > (unsigned)m) throw ParseError("scaling_list_pred_matrix_id_delta out of range");
        if (delta == 0) {
          default_list(size_id, m, sl.coef[size_id][m], &sl.dc[size_id][m]);
        } else {
          const int ref = m - (int)delta * step;
          memcpy(sl.coef[size_id][m], sl.coef[size_id][ref], sizeof(int32_t) * 64);
          sl.dc[size_id][m] = sl.dc[size_id][ref];
        }
      } else {
        int next = 8;
        sl.dc[size_id][m] = 16;
        if (size_id > 1) {
          const int d = br.se();
          if (d < -7 || d > 247) throw ParseError("scaling_list_dc_coef_minus8 out of range");
          next = d + 8;
          sl.dc[size_id][m] = next;
        }
        const int* scan = size_id == 0 ? scan4 : scan8;
        for (int i = 0; i < n; i++) {
          const int d = br.se();
          if (d < -128 || d > 127) throw ParseError("scaling_list_delta_coef out of range");
          next = (next + d + 256) % 256;
          sl.coef[size_id][m][scan[i]] = next;
        }
      }
    }
  }
}

// 7.3.3 profile_tier_level(1, maxNumSubLayersMinus1); returns general_profile_idc
static int parse_ptl(BitReader& br, int max_sub_layers_minus1) {
  br.u(2);
  br.u(1);
  const int profile = br.u(5);
  br.skip(32 + 4 + 43 + 1);
  br.u(8);
  bool sub_profile[8] = {false}, sub_level[8] = {false};
  for (int i = 0; i < max_sub_layers_minus1; i++) { sub_profile[i] = br.flag(); sub_level[i] = br.flag(); }
  if (max_sub_layers_minus1 > 0) for (int i = max_sub_layers_minus1; i < 8; i++) br.u(2);
  for (int i = 0; i < max_sub_layers_minus1; i++) {
    if (sub_profile[i]) br.skip(88);
    if (sub_level[i]) br.skip(8);
  }
  return profile;
}

// 7.3.2.1
std::shared_ptr<Vps> parse_vps(BitReader& br) {
  auto v = std::make_shared<Vps>();
  v->id = br.u(4);               // the rest (layer sets, HRD) does not influence decoding of a single layer
  return v;
}

// 7.3.7 st_ref_pic_set(stRpsIdx)
static void parse_st_rps(BitReader& br, int idx, int num_in_sps, const std::vector<ShortTermRps>& sets, ShortTermRps& out) {
  out = ShortTermRps();
  const bool inter = idx != 0 && br.flag();
  if (inter) {
    const int delta_idx = idx == num_in_sps ? (int)br.ue() + 1 : 1;
    if (delta_idx > idx) throw ParseError("delta_idx_minus1 out of range");
    const ShortTermRps& ref = sets[idx - delta_idx];
    const bool sign = br.flag();
    const int delta_rps = (1 - 2 * (int)sign) * ((int)br.ue() + 1);
    const int nref = ref.num_delta_pocs();
    bool used[17], use_delta[17];
    for (int j = 0; j <= nref; j++) {
      used[j] = br.flag();
      use_delta[j] = used[j] ? true : br.flag();
    }
    const int* s0 = ref.delta_poc;                       // negative part of the reference set
    const int* s1 = ref.delta_poc + ref.num_negative;    // positive part
    int neg[16], pos[16], nn = 0, np = 0;
    bool uneg[16], upos[16];
    auto push = [](int* d, bool* u, int& n, int v, bool f) { if (n >= 16) throw ParseError("short-term RPS too large"); d[n] = v; u[n++] = f; };
    for (int j = ref.num_positive - 1; j >= 0; j--) {
      const int d = s1[j] + delta_rps;
      if (d < 0 && use_delta[ref.num_negative + j]) push(neg, uneg, nn, d, used[ref.num_negative + j]);
    }
    if (delta_rps < 0 && use_delta[nref]) push(neg, uneg, nn, delta_rps, used[nref]);
    for (int j = 0; j < ref.num_negative; j++) {
      const int d = s0[j] + delta_rps;
      if (d < 0 && use_delta[j]) push(neg, uneg, nn, d, used[j]);
    }
    for (int j = ref.num_negative - 1; j >= 0; j--) {
      const int d = s0[j] + delta_rps;
      if (d > 0 && use_delta[j]) push(pos, upos, np, d, used[j]);
    }
    if (delta_rps > 0 && use_delta[nref]) push(pos, upos, np, delta_rps, used[nref]);
    for (int j = 0; j < ref.num_positive; j++) {
      const int d = s1[j] + delta_rps;
      if (d > 0 && use_delta[ref.num_negative + j]) push(pos, upos, np, d, used[ref.num_negative + j]);
    }
    if (nn + np > 16) throw ParseError("short-term RPS too large");
    out.num_negative = nn;
    out.num_positive = np;
    for (int i = 0; i < nn; i++) { out.delta_poc[i] = neg[i]; out.used[i] = uneg[i]; }
    for (int i = 0; i < np; i++) { out.delta_poc[nn + i] = pos[i]; out.used[nn + i] = upos[i]; }
  } else {
    const unsigned nn = br.ue(), np = br.ue();
    if (nn + np > 16) throw ParseError("short-term RPS too large");
    out.num_negative = nn;
    out.num_positive = np;
    int poc = 0;
    for (unsigned i = 0; i < nn; i++) { poc -= (int)br.ue() + 1; out.delta_poc[i] = poc; out.used[i] = br.flag(); }
    poc = 0;
    for (unsigned i = 0; i < np; i++) { poc += (int)br.ue() + 1; out.delta_poc[nn + i] = poc; out.used[nn + i] = br.flag(); }
  }
}

// E.2.2 / E.2.3 hrd_parameters(): parsed only to get past it
static void skip_sub_layer_hrd(BitReader& br, int cpb_cnt, bool sub_pic) {
  for (int i = 0; i < cpb_cnt; i++) {
    br.ue(); br.ue();
    if (sub_pic) { br.ue(); br.ue(); }
    br.u(1);
  }
}
static void skip_hrd(BitReader& br, bool common, int max_sub_layers_minus1) {
  bool nal = false, vcl = false, sub_pic = false;
  if (common) {
    nal = br.flag();
    vcl = br.flag();
    if (nal || vcl) {
      sub_pic = br.flag();
      if (sub_pic) { br.u(8); br.u(5); br.u(1); br.u(5); }
      br.u(4); br.u(4);
      if (sub_pic) br.u(4);
      br.u(5); br.u(5); br.u(5);
    }
  }
  for (int i = 0; i <= max_sub_layers_minus1; i++) {
    const bool fixed_general = br.flag();
    bool fixed_within_cvs = true, low_delay = false;
    if (!fixed_general) fixed_within_cvs = br.flag();
    if (fixed_within_cvs) br.ue(); else low_delay = br.flag();
    int cpb_cnt = 1;
    if (!low_delay) cpb_cnt = (int)br.ue() + 1;
    if (nal) skip_sub_layer_hrd(br, cpb_cnt, sub_pic);
    if (vcl) skip_sub_layer_hrd(br, cpb_cnt, sub_pic);
  }
}

// E.2.1 vui_parameters(): nothing of it is needed for reconstruction
static void skip_vui(BitReader& br, int max_sub_layers_minus1) {
  if (br.flag()) { if (br.u(8) == 255) { br.u(16); br.u(16); } }
  if (br.flag()) br.u(1);
  if (br.flag()) { br.u(3); br.u(1); if (br.flag()) { br.u(8); br.u(8); br.u(8); } }
  if (br.flag()) { br.ue(); br.ue(); }
  br.u(1); br.u(1); br.u(1);
  if (br.flag()) { br.ue(); br.ue(); br.ue(); br.ue(); }
  if (br.flag()) {
    br.u(32); br.u(32);
    if (br.flag()) br.ue();
    if (br.flag()) skip_hrd(br, true, max_sub_layers_minus1);
  }
  if (br.flag()) { br.u(1); br.u(1); br.u(1); br.ue(); br.ue(); br.ue(); br.ue(); br.ue(); }
}

// 7.3.2.2
std::shared_ptr<Sps> parse_sps(BitReader& br) {
  auto sp = std::make_shared<Sps>();
  Sps& s = *sp;
  s.vps_id = br.u(4);
  const int msl1 = br.u(3);
  s.max_sub_layers = msl1 + 1;
  br.u(1);
  const int profile = parse_ptl(br, msl1);
  (void)profile;
  s.id = br.ue();
  if (s.id > 15) throw ParseError("sps_seq_parameter_set_id out of range");
  s.chroma_format_idc = br.ue();
  if (s.chroma_format_idc > 3) throw ParseError("chroma_format_idc out of range");
  if (s.chroma_format_idc == 3 && br.u(1)) throw Unsupported("separate_colour_plane_flag (three monochrome pictures per access unit)");
  s.width = br.ue();
  s.height = br.ue();
  if (br.flag()) {                       // conformance window, in units of SubWidthC / SubHeightC (Table 6-1): 2 / 2 for 4:2:0, 2 / 1 for 4:2:2, else 1 / 1
    const int ux = (s.chroma_format_idc == 1 || s.chroma_format_idc == 2) ? 2 : 1, uy = s.chroma_format_idc == 1 ? 2 : 1;
    s.conf_left = ux * br.ue(); s.conf_right = ux * br.ue(); s.conf_top = uy * br.ue(); s.conf_bottom = uy * br.ue();
  }
  s.bit_depth_luma = 8 + br.ue();
  s.bit_depth_chroma = 8 + br.ue();
  if (s.bit_depth_luma > 12 || s.bit_depth_chroma > 12) throw Unsupported("bit depths above 12 are outside the device path");
  s.log2_max_poc_lsb = 4 + br.ue();
  if (s.log2_max_poc_lsb > 16) throw ParseError("log2_max_pic_order_cnt_lsb_minus4 out of range");
  const bool sub_layer_ordering = br.flag();
  for (int i = sub_layer_ordering ? 0 : msl1; i <= msl1; i++) {
    s.max_dec_pic_buffering[i] = br.ue() + 1;
    s.num_reorder_pics[i] = br.ue();
    s.max_latency_increase_plus1[i] = br.ue();
  }
  if (!sub_layer_ordering)
    for (int i = 0; i < msl1; i++) {
      s.max_dec_pic_buffering[i] = s.max_dec_pic_buffering[msl1];
      s.num_reorder_pics[i] = s.num_reorder_pics[msl1];
      s.max_latency_increase_plus1[i] = s.max_latency_increase_plus1[msl1];
    }
  s.log2_min_cb = 3 + br.ue();
  s.log2_ctb = s.log2_min_cb + br.ue();
  s.log2_min_tb = 2 + br.ue();
  s.log2_max_tb = s.log2_min_tb + br.ue();
  s.max_th_depth_inter = br.ue();
  s.max_th_depth_intra = br.ue();
  if (s.log2_ctb < 4 || s.log2_ctb > 6 || s.log2_max_tb > 5 || s.log2_max_tb > s.log2_ctb || s.log2_min_tb >= s.log2_min_cb)
    throw ParseError("coding block / transform block sizes out of range");
  if (s.width <= 0 || s.height <= 0 || (s.width & ((1 << s.log2_min_cb) - 1)) || (s.height & ((1 << s.log2_min_cb) - 1)))
    throw ParseError("picture size is not a multiple of the minimum coding block");
  s.scaling_list_enabled = br.flag();
  if (s.scaling_list_enabled) {
    s.scaling_lists.set_default();
    s.sps_scaling_list_data_present = br.flag();
    if (s.sps_scaling_list_data_present) parse_scaling_list_data(br, s.scaling_lists);
  }
  s.amp = br.flag();
  s.sao = br.flag();
  s.pcm = br.flag();
  if (s.pcm) {
    s.pcm_bit_depth_luma = br.u(4) + 1;
    s.pcm_bit_depth_chroma = br.u(4) + 1;
    s.log2_min_pcm_cb = 3 + br.ue();
    s.log2_max_pcm_cb = s.log2_min_pcm_cb + br.ue();
    s.pcm_loop_filter_disabled = br.flag();
  }
  const unsigned nst = br.ue();
  if (nst > 64) throw ParseError("num_short_term_ref_pic_sets out of range");
  s.st_rps.resize(nst);
  for (unsigned i = 0; i < nst; i++) parse_st_rps(br, i, nst, s.st_rps, s.st_rps[i]);
  s.long_term_ref_pics_present = br.flag();
  if (s.long_term_ref_pics_present) {
    s.num_long_term_ref_pics_sps = br.ue();
    if (s.num_long_term_ref_pics_sps > 32) throw ParseError("num_long_term_ref_pics_sps out of range");
    for (int i = 0; i < s.num_long_term_ref_pics_sps; i++) {
      s.lt_ref_pic_poc_lsb_sps[i] = br.u(s.log2_max_poc_lsb);
      s.used_by_curr_pic_lt_sps[i] = br.flag();
    }
  }
  s.temporal_mvp = br.flag();
  s.strong_intra_smoothing = br.flag();
  if (br.flag()) skip_vui(br, msl1);
  if (br.flag()) {                       // sps_extension_present_flag: range extension flag first (HM 16.0: TDecCAVLC.cpp:758-800)
    const bool range_ext = br.flag();
    br.u(7);
    if (range_ext) {
      s.rext_rotation = br.flag();
      s.rext_ts_context = br.flag();
      s.rext_implicit_rdpcm = br.flag();
      s.rext_explicit_rdpcm = br.flag();
      if (br.flag()) throw Unsupported("extended_precision_processing (RExt)");
      s.rext_intra_smoothing_disabled = br.flag();
      s.rext_high_precision_offsets = br.flag();
      s.rext_persistent_rice = br.flag();
      if (br.flag()) throw Unsupported("cabac_bypass_alignment (RExt)");
    }
  }
  return sp;
}

// 6.5.1 CTB raster <-> tile scan conversion
void Pps::derive_tiles(const Sps& sps) {
  const int W = sps.pic_w_ctbs(), H = sps.pic_h_ctbs();
  std::vector<int> cw(num_tile_cols), rh(num_tile_rows);
  if (uniform_spacing) {
    for (int i = 0; i < num_tile_cols; i++) cw[i] = ((i + 1) * W) / num_tile_cols - (i * W) / num_tile_cols;
    for (int i = 0; i < num_tile_rows; i++) rh[i] = ((i + 1) * H) / num_tile_rows - (i * H) / num_tile_rows;
  } else {
    int rest = W;
    for (int i = 0; i + 1 < num_tile_cols; i++) { cw[i] = col_width_minus1[i] + 1; rest -= cw[i]; }
    cw[num_tile_cols - 1] = rest;
    rest = H;
    for (int i = 0; i + 1 < num_tile_rows; i++) { rh[i] = row_height_minus1[i] + 1; rest -= rh[i]; }
    rh[num_tile_rows - 1] = rest;
  }
  for (int v : cw) if (v <= 0) throw ParseError("tile column width out of range");
  for (int v : rh) if (v <= 0) throw ParseError("tile row height out of range");
  col_bd.assign(num_tile_cols + 1, 0);
  row_bd.assign(num_tile_rows + 1, 0);
  for (int i = 0; i < num_tile_cols; i++) col_bd[i + 1] = col_bd[i] + cw[i];
  for (int i = 0; i < num_tile_rows; i++) row_bd[i + 1] = row_bd[i] + rh[i];
  ctb_rs_to_ts.assign(W * H, 0);
  ctb_ts_to_rs.assign(W * H, 0);
  tile_id.assign(W * H, 0);
  for (int rs = 0; rs < W * H; rs++) {
    const int tbx = rs % W, tby = rs / W;
    int tx = 0, ty = 0;
    for (int i = 0; i < num_tile_cols; i++) if (tbx >= col_bd[i]) tx = i;
    for (int i = 0; i < num_tile_rows; i++) if (tby >= row_bd[i]) ty = i;
    int ts = 0;
    for (int i = 0; i < tx; i++) ts += rh[ty] * cw[i];
    for (int i = 0; i < ty; i++) ts += W * rh[i];
    ts += (tby - row_bd[ty]) * cw[tx] + tbx - col_bd[tx];
    ctb_rs_to_ts[rs] = ts;
    ctb_ts_to_rs[ts] = rs;
  }
  int tid = 0;
  for (int j = 0; j < num_tile_rows; j++)
    for (int i = 0; i < num_tile_cols; i++, tid++)
      for (int y = row_bd[j]; y < row_bd[j + 1]; y++)
        for (int x = col_bd[i]; x < col_bd[i + 1]; x++) tile_id[ctb_rs_to_ts[y * W + x]] = tid;
}

// 7.3.2.3
std::shared_ptr<Pps> parse_pps(BitReader& br) {
  auto pp = std::make_shared<Pps>();
  Pps& p = *pp;
  p.id = br.ue();
  if (p.id > 63) throw ParseError("pps_pic_parameter_set_id out of range");
  p.sps_id = br.ue();
  if (p.sps_id > 15) throw ParseError("pps_seq_parameter_set_id out of range");
  p.dependent_slice_segments_enabled = br.flag();
  p.output_flag_present = br.flag();
  p.num_extra_slice_header_bits = br.u(3);
  p.sign_data_hiding = br.flag();
  p.cabac_init_present = br.flag();
  p.num_ref_idx_default[0] = br.ue() + 1;
  p.num_ref_idx_default[1] = br.ue() + 1;
  if (p.num_ref_idx_default[0] > 15 || p.num_ref_idx_default[1] > 15) throw ParseError("num_ref_idx_default_active out of range");
  p.init_qp = 26 + br.se();
  p.constrained_intra_pred = br.flag();
  p.transform_skip_enabled = br.flag();
  p.cu_qp_delta_enabled = br.flag();
  if (p.cu_qp_delta_enabled) p.diff_cu_qp_delta_depth = br.ue();
  p.cb_qp_offset = br.se();
  p.cr_qp_offset = br.se();
  if (p.cb_qp_offset < -12 || p.cb_qp_offset > 12 || p.cr_qp_offset < -12 || p.cr_qp_offset > 12) throw ParseError("pps chroma QP offset out of range");
  p.slice_chroma_qp_offsets_present = br.flag();
  p.weighted_pred = br.flag();
  p.weighted_bipred = br.flag();
  p.transquant_bypass_enabled = br.flag();
  p.tiles_enabled = br.flag();
  p.entropy_coding_sync = br.flag();
  if (p.tiles_enabled) {
    p.num_tile_cols = br.ue() + 1;
    p.num_tile_rows = br.ue() + 1;
    if (p.num_tile_cols > 20 || p.num_tile_rows > 22) throw ParseError("tile grid out of range");
    p.uniform_spacing = br.flag();
    if (!p.uniform_spacing) {
      for (int i = 0; i + 1 < p.num_tile_cols; i++) p.col_width_minus1.push_back(br.ue());
      for (int i = 0; i + 1 < p.num_tile_rows; i++) p.row_height_minus1.push_back(br.ue());
    }
    p.lf_across_tiles = br.flag();
  }
  p.lf_across_slices = br.flag();
  p.deblocking_control_present = br.flag();
  if (p.deblocking_control_present) {
    p.deblocking_override_enabled = br.flag();
    p.deblocking_disabled = br.flag();
    if (!p.deblocking_disabled) { p.beta_offset_div2 = br.se(); p.tc_offset_div2 = br.se(); }
  }
  p.scaling_list_data_present = br.flag();
  if (p.scaling_list_data_present) { p.scaling_lists.set_default(); parse_scaling_list_data(br, p.scaling_lists); }
  p.lists_modification_present = br.flag();
  p.log2_par_mrg_level = 2 + br.ue();
  p.slice_header_extension_present = br.flag();
  if (br.flag()) {                       // pps_extension_present_flag (HM 16.0: TDecCAVLC.cpp:322-392)
    const bool range_ext = br.flag();
    br.u(7);
    if (range_ext) {
      if (p.transform_skip_enabled) {
        p.log2_max_ts_size = 2 + (int)br.ue();
        if (p.log2_max_ts_size > 5) throw ParseError("log2_max_transform_skip_block_size_minus2 out of range");
      }
      p.cross_component_prediction = br.flag();           // (only meaningful with ChromaArrayType 3: checked where the PPS is activated)
      if (br.flag()) throw Unsupported("CU-level chroma QP offsets (RExt)");
      for (int k = 0; k < 2; k++) {
        p.sao_offset_shift[k] = (int)br.ue();
        if (p.sao_offset_shift[k] > 5) throw ParseError("log2_sao_offset_scale out of range");   // at most BitDepth - 10 in a conforming stream
      }
    }
  }
  return pp;
}

// 7.3.6.3 pred_weight_table()
static void parse_pred_weight_table(BitReader& br, const Sps& sps, SliceHeader& sh) {
  // high_precision_offsets_enabled_flag: offsets in units of the coding bit depth (HM 16.0: TDecCAVLC.cpp:1851-1879)
  const int lrange = sps.rext_high_precision_offsets ? 1 << (sps.bit_depth_luma - 1) : 128, crange = sps.rext_high_precision_offsets ? 1 << (sps.bit_depth_chroma - 1) : 128;
  sh.luma_log2_weight_denom = br.ue();
  if (sh.luma_log2_weight_denom > 7) throw ParseError("luma_log2_weight_denom out of range");
  const bool chroma = sps.chroma_format_idc != 0;         // monochrome: no chroma syntax (HM 16.0: TDecCAVLC.cpp:1801, 1827, 1861)
  sh.chroma_log2_weight_denom = sh.luma_log2_weight_denom + (chroma ? br.se() : 0);
  if (sh.chroma_log2_weight_denom < 0 || sh.chroma_log2_weight_denom > 7) throw ParseError("chroma log2 weight denominator out of range");
  for (int l = 0; l < (sh.type == SLICE_B ? 2 : 1); l++) {
    const int n = sh.num_ref_idx[l];
    for (int i = 0; i < n; i++) { sh.pw[l][i] = PredWeight(); sh.pw[l][i].luma_flag = br.flag(); }
    if (chroma) for (int i = 0; i < n; i++) sh.pw[l][i].chroma_flag = br.flag();
    for (int i = 0; i < n; i++) {
      PredWeight& w = sh.pw[l][i];
      w.luma_weight = 1 << sh.luma_log2_weight_denom;
      w.chroma_weight[0] = w.chroma_weight[1] = 1 << sh.chroma_log2_weight_denom;
      if (w.luma_flag) {
        const int d = br.se();
        if (d < -128 || d > 127) throw ParseError("delta_luma_weight out of range");
        w.luma_weight += d;
        w.luma_offset = br.se();
        if (w.luma_offset < -lrange || w.luma_offset >= lrange) throw ParseError("luma_offset out of range");
      }
      if (w.chroma_flag)
        for (int j = 0; j < 2; j++) {
          const int d = br.se();
          if (d < -128 || d > 127) throw ParseError("delta_chroma_weight out of range");
          w.chroma_weight[j] += d;
          const int o = br.se();
          if (o < -4 * crange || o >= 4 * crange) throw ParseError("delta_chroma_offset out of range");
          const int v = o - ((crange * w.chroma_weight[j]) >> sh.chroma_log2_weight_denom) + crange;      // (7-56)
          w.chroma_offset[j] = std::min(crange - 1, std::max(-crange, v));
        }
    }
  }
}

// 7.3.6.1 slice_segment_header()
void parse_slice_header(BitReader& br, int nal_type, int temporal_id, const ParamSets& ps, const SliceHeader* prev, SliceHeader& sh) {
  const bool first = br.flag();
  bool no_output_prior = false;
  if (nal_type >= NAL_BLA_W_LP && nal_type <= NAL_RSV_IRAP_VCL23) no_output_prior = br.flag();
  const unsigned pps_id = br.ue();
  if (pps_id > 63 || !ps.pps[pps_id]) throw ParseError("slice refers to a PPS that was not received");
  const Pps& pps = *ps.pps[pps_id];
  if (!ps.sps[pps.sps_id]) throw ParseError("PPS refers to an SPS that was not received");
  const Sps& sps = *ps.sps[pps.sps_id];
  bool dependent = false;
  int address = 0;
  if (!first) {
    if (pps.dependent_slice_segments_enabled) dependent = br.flag();
    address = br.u(ceil_log2(sps.num_ctbs()));
    if (address >= sps.num_ctbs()) throw ParseError("slice_segment_address out of range");
  }
  if (dependent) {
    if (!prev) throw ParseError("dependent slice segment without a preceding independent one");
    sh = *prev;                          // 7.4.7.1: everything else is inferred from the preceding independent segment
  } else {
    sh = SliceHeader();
  }
  sh.nal_type = nal_type;
  sh.temporal_id = temporal_id;
  sh.first_slice_segment_in_pic = first;
  sh.no_output_of_prior_pics = no_output_prior;
  sh.pps_id = pps_id;
  sh.dependent = dependent;
  sh.segment_address = address;
  sh.entry_points.clear();
  if (!dependent) {
    sh.slice_address = address;
    br.skip(pps.num_extra_slice_header_bits);
    const unsigned type = br.ue();
    if (type > 2) throw ParseError("slice_type out of range");
    sh.type = type;
    if (nal_type >= NAL_BLA_W_LP && nal_type <= NAL_RSV_IRAP_VCL23 && type != SLICE_I) throw ParseError("IRAP picture with a P or B slice");
    sh.pic_output = pps.output_flag_present ? br.flag() : true;
    const bool idr = nal_type == NAL_IDR_W_RADL || nal_type == NAL_IDR_N_LP;
    sh.temporal_mvp = false;
    if (!idr) {
      sh.poc_lsb = br.u(sps.log2_max_poc_lsb);
      if (!br.flag()) {
        parse_st_rps(br, (int)sps.st_rps.size(), (int)sps.st_rps.size(), sps.st_rps, sh.rps);
      } else {
        const int n = (int)sps.st_rps.size();
        if (n == 0) throw ParseError("short_term_ref_pic_set_sps_flag without SPS sets");
        const int idx = n > 1 ? br.u(ceil_log2(n)) : 0;
        if (idx >= n) throw ParseError("short_term_ref_pic_set_idx out of range");
        sh.rps = sps.st_rps[idx];
      }
      if (sps.long_term_ref_pics_present) {
        const int num_sps = sps.num_long_term_ref_pics_sps > 0 ? (int)br.ue() : 0;
        const int num_pics = br.ue();
        if (num_sps > sps.num_long_term_ref_pics_sps || num_sps + num_pics > 32) throw ParseError("too many long-term pictures");
        sh.num_long_term = num_sps + num_pics;
        int cycle = 0;
        for (int i = 0; i < sh.num_long_term; i++) {
          if (i < num_sps) {
            const int k = sps.num_long_term_ref_pics_sps > 1 ? br.u(ceil_log2(sps.num_long_term_ref_pics_sps)) : 0;
            if (k >= sps.num_long_term_ref_pics_sps) throw ParseError("lt_idx_sps out of range");
            sh.lt_poc[i] = sps.lt_ref_pic_poc_lsb_sps[k];
            sh.lt_used[i] = sps.used_by_curr_pic_lt_sps[k];
          } else {
            sh.lt_poc[i] = br.u(sps.log2_max_poc_lsb);
            sh.lt_used[i] = br.flag();
          }
          sh.lt_msb_present[i] = br.flag();
          if (sh.lt_msb_present[i]) {
            const int d = br.ue();
            cycle = (i == 0 || i == num_sps) ? d : cycle + d;         // (7-52)
            sh.lt_msb_cycle[i] = cycle;                                // DeltaPocMsbCycleLt: applied once the POC is known
          }
        }
      }
      if (sps.temporal_mvp) sh.temporal_mvp = br.flag();
    } else {
      sh.poc_lsb = 0;
    }
    sh.num_pic_total_curr = 0;
    for (int i = 0; i < sh.rps.num_delta_pocs(); i++) sh.num_pic_total_curr += sh.rps.used[i];
    for (int i = 0; i < sh.num_long_term; i++) sh.num_pic_total_curr += sh.lt_used[i];
    if (sps.sao) { sh.sao_luma = br.flag(); sh.sao_chroma = sps.chroma_format_idc != 0 && br.flag(); }     // (:1170: absent for 4:0:0)
    sh.num_ref_idx[0] = sh.num_ref_idx[1] = 0;
    sh.collocated_from_l0 = true;
    sh.collocated_ref_idx = 0;
    if (type != SLICE_I) {
      sh.num_ref_idx[0] = pps.num_ref_idx_default[0];
      sh.num_ref_idx[1] = type == SLICE_B ? pps.num_ref_idx_default[1] : 0;
      if (br.flag()) {
        sh.num_ref_idx[0] = br.ue() + 1;
        if (type == SLICE_B) sh.num_ref_idx[1] = br.ue() + 1;
      }
      if (sh.num_ref_idx[0] > 15 || sh.num_ref_idx[1] > 15) throw ParseError("num_ref_idx_active out of range");
      if (sh.num_pic_total_curr == 0) throw ParseError("P or B slice without reference pictures");
      if (pps.lists_modification_present && sh.num_pic_total_curr > 1) {       // 7.3.6.2
        const int bits = ceil_log2(sh.num_pic_total_curr);
        for (int l = 0; l < (type == SLICE_B ? 2 : 1); l++) {
          sh.list_mod_flag[l] = br.flag();
          if (sh.list_mod_flag[l])
            for (int i = 0; i < sh.num_ref_idx[l]; i++) {
              sh.list_entry[l][i] = br.u(bits);
              if (sh.list_entry[l][i] >= sh.num_pic_total_curr) throw ParseError("list_entry out of range");
            }
        }
      }
      if (type == SLICE_B) sh.mvd_l1_zero = br.flag();
      if (pps.cabac_init_present) sh.cabac_init_flag = br.flag();
      if (sh.temporal_mvp) {
        if (type == SLICE_B) sh.collocated_from_l0 = br.flag();
        if ((sh.collocated_from_l0 && sh.num_ref_idx[0] > 1) || (!sh.collocated_from_l0 && sh.num_ref_idx[1] > 1)) {
          sh.collocated_ref_idx = br.ue();
          if (sh.collocated_ref_idx >= sh.num_ref_idx[sh.collocated_from_l0 ? 0 : 1]) throw ParseError("collocated_ref_idx out of range");
        }
      }
      if ((pps.weighted_pred && type == SLICE_P) || (pps.weighted_bipred && type == SLICE_B)) parse_pred_weight_table(br, sps, sh);
      const unsigned five_minus = br.ue();
      if (five_minus > 4) throw ParseError("five_minus_max_num_merge_cand out of range");
      sh.max_num_merge_cand = 5 - five_minus;
    }
    sh.qp = pps.init_qp + br.se();
    if (sh.qp < -6 * (sps.bit_depth_luma - 8) || sh.qp > 51) throw ParseError("slice QP out of range");
    sh.cb_qp_offset = sh.cr_qp_offset = 0;
    if (pps.slice_chroma_qp_offsets_present && sps.chroma_format_idc != 0) {       // (HM 16.0 reads them per valid component: TDecCAVLC.cpp:1349-1368)
      sh.cb_qp_offset = br.se();
      sh.cr_qp_offset = br.se();
      if (sh.cb_qp_offset < -12 || sh.cb_qp_offset > 12 || sh.cr_qp_offset < -12 || sh.cr_qp_offset > 12) throw ParseError("slice chroma QP offset out of range");
    }
    bool override_flag = false;
    if (pps.deblocking_override_enabled) override_flag = br.flag();
    sh.deblocking_disabled = pps.deblocking_disabled;
    sh.beta_offset_div2 = pps.beta_offset_div2;
    sh.tc_offset_div2 = pps.tc_offset_div2;
    if (override_flag) {
      sh.deblocking_disabled = br.flag();
      if (!sh.deblocking_disabled) { sh.beta_offset_div2 = br.se(); sh.tc_offset_div2 = br.se(); }
    }
    if (sh.beta_offset_div2 < -6 || sh.beta_offset_div2 > 6 || sh.tc_offset_div2 < -6 || sh.tc_offset_div2 > 6) throw ParseError("deblocking offsets out of range");
    sh.lf_across_slices = pps.lf_across_slices;
    if (pps.lf_across_slices && (sh.sao_luma || sh.sao_chroma || !sh.deblocking_disabled)) sh.lf_across_slices = br.flag();
  }
  if (pps.tiles_enabled || pps.entropy_coding_sync) {
    const unsigned n = br.ue();
    if (n > (unsigned)sps.num_ctbs()) throw ParseError("num_entry_point_offsets out of range");
    if (n > 0) {
      const int len = br.ue() + 1;
      if (len > 32) throw ParseError("offset_len_minus1 out of range");
      for (unsigned i = 0; i < n; i++) sh.entry_points.push_back(br.u(len) + 1);
    }
  }
  if (pps.slice_header_extension_present) {
    const unsigned n = br.ue();
    br.skip(8 * (size_t)n);
  }
  br.byte_alignment();
  sh.data_bit_offset = br.pos();
}

}  // namespace hmdec
