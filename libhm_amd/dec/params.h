// params.h -- parameter sets and slice segment header of the host-side HEVC parser.
// Rec. ITU-T H.265 (04/2013 + RExt flags as HM 16.0 reads them) 7.3.2.1-7.3.2.3, 7.3.4, 7.3.6, 7.3.7, E.2.
// HM counterpart: TDecCAVLC.cpp parseVPS/parseSPS/parsePPS/parseSliceHeader (:169-1560).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <vector>

#include "bitreader.h"

namespace hmdec {

enum NalType {
  NAL_TRAIL_N = 0, NAL_TRAIL_R = 1, NAL_TSA_N = 2, NAL_TSA_R = 3, NAL_STSA_N = 4, NAL_STSA_R = 5, NAL_RADL_N = 6, NAL_RADL_R = 7,
  NAL_RASL_N = 8, NAL_RASL_R = 9, NAL_RSV_VCL_N14 = 14, NAL_BLA_W_LP = 16, NAL_BLA_W_RADL = 17, NAL_BLA_N_LP = 18, NAL_IDR_W_RADL = 19,
  NAL_IDR_N_LP = 20, NAL_CRA = 21, NAL_RSV_IRAP_VCL23 = 23, NAL_RSV_VCL31 = 31, NAL_VPS = 32, NAL_SPS = 33, NAL_PPS = 34, NAL_AUD = 35,
  NAL_EOS = 36, NAL_EOB = 37, NAL_FD = 38, NAL_PREFIX_SEI = 39, NAL_SUFFIX_SEI = 40
};
enum SliceType { SLICE_B = 0, SLICE_P = 1, SLICE_I = 2 };

struct ScalingListSet {             // 7.3.4: ScalingFactor source lists, raster order (16 values for 4x4, 64 otherwise)
  int32_t coef[4][6][64];
  int32_t dc[4][6];
  void set_default();               // Tables 7-5 / 7-6
};

struct ShortTermRps {               // 7.4.8, after inter RPS prediction: delta POCs, negative first (closest first), then positive
  int num_negative = 0, num_positive = 0;
  int delta_poc[16] = {0};          // [0, num_negative): S0, then S1
  bool used[16] = {false};
  int num_delta_pocs() const { return num_negative + num_positive; }
};

struct Vps { int id = 0; };

struct Sps {
  int id = 0, vps_id = 0, max_sub_layers = 1;
  int chroma_format_idc = 1, width = 0, height = 0;
  // chroma subsampling (Table 6-1): log2 SubWidthC / SubHeightC; monochrome keeps 4:2:0-shaped dummy planes
  int csx() const { return chroma_format_idc == 3 ? 0 : 1; }
  int csy() const { return (chroma_format_idc == 2 || chroma_format_idc == 3) ? 0 : 1; }
  int conf_left = 0, conf_right = 0, conf_top = 0, conf_bottom = 0;       // in luma samples
  int bit_depth_luma = 8, bit_depth_chroma = 8, log2_max_poc_lsb = 4;
  int max_dec_pic_buffering[8] = {0}, num_reorder_pics[8] = {0}, max_latency_increase_plus1[8] = {0};
  int log2_min_cb = 3, log2_ctb = 6, log2_min_tb = 2, log2_max_tb = 5, max_th_depth_inter = 0, max_th_depth_intra = 0;
  bool scaling_list_enabled = false, sps_scaling_list_data_present = false;
  ScalingListSet scaling_lists;
  bool amp = false, sao = false, pcm = false, pcm_loop_filter_disabled = false;
  int pcm_bit_depth_luma = 8, pcm_bit_depth_chroma = 8, log2_min_pcm_cb = 3, log2_max_pcm_cb = 5;
  std::vector<ShortTermRps> st_rps;
  bool long_term_ref_pics_present = false;
  int num_long_term_ref_pics_sps = 0, lt_ref_pic_poc_lsb_sps[32] = {0};
  bool used_by_curr_pic_lt_sps[32] = {false};
  bool temporal_mvp = false, strong_intra_smoothing = false;
  // sps_range_extension() (HM 16.0: TDecCAVLC.cpp:778-786); tools the device path lacks are refused while parsing
  bool rext_rotation = false, rext_ts_context = false, rext_implicit_rdpcm = false, rext_explicit_rdpcm = false;
  bool rext_persistent_rice = false, rext_intra_smoothing_disabled = false, rext_high_precision_offsets = false;
  int range_ext_flags() const {                // HMGPU_REXT_*
    return (rext_rotation ? 1 : 0) | (rext_implicit_rdpcm ? 2 : 0) | (rext_explicit_rdpcm ? 4 : 0) | (rext_intra_smoothing_disabled ? 8 : 0);
  }
  // derived
  int ctb_size() const { return 1 << log2_ctb; }
  int pic_w_ctbs() const { return (width + ctb_size() - 1) >> log2_ctb; }
  int pic_h_ctbs() const { return (height + ctb_size() - 1) >> log2_ctb; }
  int num_ctbs() const { return pic_w_ctbs() * pic_h_ctbs(); }
};

struct Pps {
  int id = 0, sps_id = 0;
  bool dependent_slice_segments_enabled = false, output_flag_present = false, sign_data_hiding = false, cabac_init_present = false;
  int num_extra_slice_header_bits = 0, num_ref_idx_default[2] = {1, 1}, init_qp = 26;
  bool constrained_intra_pred = false, transform_skip_enabled = false, cu_qp_delta_enabled = false;
  int diff_cu_qp_delta_depth = 0, cb_qp_offset = 0, cr_qp_offset = 0;
  bool slice_chroma_qp_offsets_present = false, weighted_pred = false, weighted_bipred = false, transquant_bypass_enabled = false;
  bool tiles_enabled = false, entropy_coding_sync = false, uniform_spacing = true, lf_across_tiles = true;
  bool cross_component_prediction = false;         // pps_range_extension(): cross_component_prediction_enabled_flag (4:4:4)
  int num_tile_cols = 1, num_tile_rows = 1;
  std::vector<int> col_width_minus1, row_height_minus1;      // explicit spacing, all but the last
  bool lf_across_slices = false, deblocking_control_present = false, deblocking_override_enabled = false, deblocking_disabled = false;
  int beta_offset_div2 = 0, tc_offset_div2 = 0;
  bool scaling_list_data_present = false;
  ScalingListSet scaling_lists;
  bool lists_modification_present = false, slice_header_extension_present = false;
  int log2_par_mrg_level = 2;
  int log2_max_ts_size = 2, sao_offset_shift[2] = {0, 0};      // pps_range_extension(): log2 MaxTbSkipSize, log2_sao_offset_scale_{luma,chroma}
  // derived once the SPS is known (6.5.1)
  std::vector<int> col_bd, row_bd, ctb_rs_to_ts, ctb_ts_to_rs, tile_id;     // tile_id indexed by TS address
  int derived_w = -1, derived_h = -1;      // picture size in CTBs the tables above were derived for
  void derive_tiles(const Sps& sps);
};

struct PredWeight { bool luma_flag = false, chroma_flag = false; int luma_weight = 0, luma_offset = 0, chroma_weight[2] = {0, 0}, chroma_offset[2] = {0, 0}; };

struct SliceHeader {
  int nal_type = 0, temporal_id = 0;
  bool first_slice_segment_in_pic = true, no_output_of_prior_pics = false, dependent = false;
  int pps_id = 0, segment_address = 0;       // CTB raster address of the first CTB of the segment
  int slice_address = 0;                     // SliceAddrRs: segment_address of the independent segment this one belongs to
  int type = SLICE_I;
  bool pic_output = true;
  int poc_lsb = 0, poc = 0;
  ShortTermRps rps;                          // the short-term RPS in force
  int num_long_term = 0, lt_poc[32] = {0}, lt_msb_cycle[32] = {0};   // PocLsbLt and DeltaPocMsbCycleLt
  bool lt_msb_present[32] = {false}, lt_used[32] = {false};
  bool temporal_mvp = false, sao_luma = false, sao_chroma = false;
  int num_ref_idx[2] = {0, 0};
  bool list_mod_flag[2] = {false, false};
  int list_entry[2][16] = {{0}};
  bool mvd_l1_zero = false, cabac_init_flag = false, collocated_from_l0 = true;
  int collocated_ref_idx = 0;
  int luma_log2_weight_denom = 0, chroma_log2_weight_denom = 0;
  PredWeight pw[2][16];
  int max_num_merge_cand = 5;
  int qp = 26, cb_qp_offset = 0, cr_qp_offset = 0;       // SliceQpY and the slice-level offsets
  bool deblocking_disabled = false, lf_across_slices = false;
  int beta_offset_div2 = 0, tc_offset_div2 = 0;
  std::vector<uint32_t> entry_points;
  size_t data_bit_offset = 0;                // first bit of slice_segment_data() in the RBSP (byte aligned)
  int num_pic_total_curr = 0;
};

struct ParamSets {
  std::shared_ptr<Vps> vps[16];
  std::shared_ptr<Sps> sps[16];
  std::shared_ptr<Pps> pps[64];
};

std::shared_ptr<Vps> parse_vps(BitReader& br);
std::shared_ptr<Sps> parse_sps(BitReader& br);
std::shared_ptr<Pps> parse_pps(BitReader& br);
// parses a slice segment header; `prev` = header of the preceding independent segment of the same picture (for dependent ones)
void parse_slice_header(BitReader& br, int nal_type, int temporal_id, const ParamSets& ps, const SliceHeader* prev, SliceHeader& sh);

}  // namespace hmdec
