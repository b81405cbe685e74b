"""The C restatement on the syntax variants HM's own decoder cannot be run on (fixtures lite_*.npz: bitstream + HM's ENCODER
reconstruction): the host parser (libhmdec, parse-only) supplies the per-CTU data, oracle/hm_oracle.c reconstructs, deblocks and
applies SAO, and the result must be the encoder's reconstruction.  This is what pins the oracle's slice-boundary rules (deblocking
and SAO across slices with slice_loop_filter_across_slices_enabled_flag 0 and 1, intra availability across slices and dependent
slice segments, slices of tiles) against data that HM itself produced -- CPU only."""
import ctypes as C

import numpy as np
import pytest

from libhm_amd import abi, hmdec
from tests import golden_util as gu

FROM_PARSER = {"depth": "depth", "part_size": "part_size", "pred_mode": "pred_mode", "qp": "qp", "tr_idx": "tr_idx", "cbf_y": "cbf0",
               "cbf_u": "cbf1", "cbf_v": "cbf2", "ts_y": "ts0", "ts_u": "ts1", "ts_v": "ts2", "mv0": "mv0", "mv1": "mv1",
               "ref_idx0": "ref_idx0", "ref_idx1": "ref_idx1", "intra_dir_l": "intra_dir0", "intra_dir_c": "intra_dir1",
               "bypass": "bypass", "ipcm": "ipcm", "slice_idx": "slice_idx", "tile_idx": "tile_idx"}


def _parsed_pictures(stream):
    """pictures in decoding order as the oracle wants them"""
    pics = []
    with hmdec.Decoder(parse_only=True) as d:
        def on_decoded(p):
            g = p.geometry()
            arrays = {k: p.array(v) for k, v in FROM_PARSER.items()}
            if g["chroma_format"] == 3:                                # cross-component prediction weights (all zero where the PPS has it off)
                arrays["ccp_u"], arrays["ccp_v"] = p.array("ccp0"), p.array("ccp1")
            slices = []
            for i in range(p.num_slices()):
                sp, lists = p.slice_params(i)
                slices.append((sp, lists))
            pics.append(dict(poc=p.poc, geom=g, arrays=arrays, coeff=[p.array("coeff%d" % c) for c in range(3)],
                             sao=p.array("sao").reshape(g["num_ctbs"], 3, 35), slices=slices, crop=p.conformance_window()))
        d.decode_stream(stream, on_decoded=on_decoded)
    return pics


@pytest.mark.parametrize("name", gu.LITE + gu.LITE_EXT + gu.SURGERY)
def test_oracle_reconstructs_the_encoders_pictures(oracle, name):
    z = gu.load("lite_" + name)
    pics = _parsed_pictures(z["bitstream"])
    assert len(pics) == int(z["geom"][2])
    index_of_poc, finals = {}, []
    multi_slice = False
    for k, p in enumerate(pics):
        g = p["geom"]
        seq = abi.make_seq(g["width"], g["height"], g["bd_y"], g["bd_c"], log2_ctu=g["log2_ctb"], max_pictures=len(pics) + 1,
                           strong_intra_smoothing=g["strong_intra"], range_ext_flags=g["range_ext"])
        seq.pcm_bit_depth_luma, seq.pcm_bit_depth_chroma, seq.pcm_loop_filter_disable = g["pcm_bd_y"], g["pcm_bd_c"], g["pcm_lf_disable"]
        seq.chroma_format = g["chroma_format"]
        sx, sy = g["csx"], g["csy"]
        parts = len(p["arrays"]["depth"]) // g["num_ctbs"]
        m = {kk: (v.reshape(g["num_ctbs"], parts, 2) if kk.startswith("mv") else v.reshape(g["num_ctbs"], -1) if v.size != g["num_ctbs"] else v)
             for kk, v in p["arrays"].items()}
        meta = abi.MetaHolder(m)
        coeffs = abi.CoeffHolder(*p["coeff"])
        slices, keep = [], []
        for sp, lists in p["slices"]:
            for l in range(2):
                for i in range(sp.num_ref_idx[l]):
                    sp.ref_pic[l][i] = index_of_poc[sp.ref_poc[l][i]]           # the oracle indexes its reference list by handle
            if sp.scaling_lists:
                keep.append(lists)
                sp.scaling_lists = C.pointer(lists)
            slices.append(sp)
        multi_slice |= len(slices) > 1
        pp = abi.make_pic_params(sao_enabled=g["sao"], lf_across_tiles=g["lf_across_tiles"], sao_offset_shift=g["sao_shift"])
        cur = [np.full((g["height"] >> (sy if c else 0), g["width"] >> (sx if c else 0)), -1, dtype=np.int16) for c in range(3)]
        oracle.decompress_ctus(seq, slices, meta, coeffs, cur, finals)
        oracle.loop_filter_pic(seq, slices, meta, pp, cur, 3)
        if g["sao"]:
            prm = oracle.sao_reconstruct_params(seq, pp, meta, p["sao"])
            cur = oracle.sao_process(seq, slices, pp, meta, prm, cur)
        l, r, t, b = p["crop"]
        for c in range(3 if ("poc%02d_1" % p["poc"]) in z else 1):        # (monochrome fixtures hold the luma plane only)
            cx, cy = (sx, sy) if c else (0, 0)
            got = cur[c][t >> cy:cur[c].shape[0] - (b >> cy), l >> cx:cur[c].shape[1] - (r >> cx)]
            assert np.array_equal(got, z["poc%02d_%d" % (p["poc"], c)]), "%s POC %d component %d" % (name, p["poc"], c)
        index_of_poc[p["poc"]] = len(finals)
        finals.append(cur)
    if "slices" in name:
        assert multi_slice
