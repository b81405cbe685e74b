"""ctypes mirror of include/hmgpu.h (the C ABI of libhmgpu.so) + helpers to fill the structs from numpy arrays.

Pure declarations: no compute here.  Field order and types must match include/hmgpu.h exactly
(tests/test_abi.py cross-checks sizes against the compiled library).
"""
import ctypes as C

import numpy as np

HMGPU_OK, HMGPU_EINVAL, HMGPU_EDEVICE, HMGPU_EUNSUPPORTED, HMGPU_ENOMEM = 0, 1, 2, 3, 4
MAX_REF = 16
NO_PIC = -1
B_SLICE, P_SLICE, I_SLICE = 0, 1, 2
MODE_INTER, MODE_INTRA = 0, 1
SIZE_2Nx2N, SIZE_2NxN, SIZE_Nx2N, SIZE_NxN, SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N, SIZE_NONE = range(9)
SAO_OFF, SAO_NEW, SAO_MERGE = 0, 1, 2
SAO_EO_0, SAO_EO_90, SAO_EO_135, SAO_EO_45, SAO_BO = range(5)
NUM_KERNELS = 12

STAGE_DEBLOCK_VER, STAGE_DEBLOCK_HOR, STAGE_SAO, STAGE_RECON = 1, 2, 4, 8


class SeqParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bit_depth_luma", C.c_int32), ("bit_depth_chroma", C.c_int32),
                ("chroma_format", C.c_int32), ("log2_ctu_size", C.c_int32), ("max_pictures", C.c_int32),
                ("pcm_loop_filter_disable", C.c_int32), ("strong_intra_smoothing", C.c_int32), ("pcm_bit_depth_luma", C.c_int32), ("pcm_bit_depth_chroma", C.c_int32),
                ("range_ext_flags", C.c_int32), ("reserved", C.c_int32 * 4)]


class ScalingLists(C.Structure):
    _fields_ = [("coef", ((C.c_int32 * 64) * 6) * 4), ("dc", (C.c_int32 * 6) * 4)]


class SliceParams(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32),
                ("pps_cb_qp_offset", C.c_int32), ("pps_cr_qp_offset", C.c_int32), ("deblocking_disable", C.c_int32),
                ("beta_offset_div2", C.c_int32), ("tc_offset_div2", C.c_int32), ("lf_across_slices", C.c_int32),
                ("weighted_pred", C.c_int32), ("lf_across_tiles", C.c_int32), ("num_ref_idx", C.c_int32 * 2),
                ("ref_pic", (C.c_int32 * MAX_REF) * 2), ("ref_poc", (C.c_int32 * MAX_REF) * 2), ("constrained_intra_pred", C.c_int32), ("reserved", C.c_int32 * 4),
                ("wp_log2_denom", C.c_int32 * 2), ("wp_weight", ((C.c_int16 * 3) * MAX_REF) * 2), ("wp_offset", ((C.c_int16 * 3) * MAX_REF) * 2),
                ("scaling_lists", C.POINTER(ScalingLists))]


class CtuMeta(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("part_size", C.c_void_p), ("pred_mode", C.c_void_p), ("qp", C.c_void_p),
                ("tr_idx", C.c_void_p), ("cbf", C.c_void_p * 3), ("transform_skip", C.c_void_p * 3),
                ("mv", C.c_void_p * 2), ("ref_idx", C.c_void_p * 2), ("intra_dir", C.c_void_p * 2),
                ("transquant_bypass", C.c_void_p), ("ipcm", C.c_void_p), ("slice_idx", C.c_void_p), ("tile_idx", C.c_void_p),
                ("ccp_alpha", C.c_void_p * 2)]


class Coeffs(C.Structure):
    _fields_ = [("level", C.c_void_p * 3), ("pcm_sample", C.c_void_p * 3), ("ctu_level_start", C.c_void_p * 3)]


class SaoParam(C.Structure):
    _fields_ = [("mode_idc", C.c_int32), ("type_idc", C.c_int32), ("type_aux_info", C.c_int32), ("offset", C.c_int32 * 32)]


class PicParams(C.Structure):
    _fields_ = [("lf_across_tiles", C.c_int32), ("sao_enabled", C.c_int32), ("sao_offset_shift_luma", C.c_int32),
                ("sao_offset_shift_chroma", C.c_int32), ("reserved", C.c_int32 * 8)]


class PictureJob(C.Structure):
    _fields_ = [("pic", C.c_int32), ("num_slices", C.c_int32), ("slices", C.POINTER(C.POINTER(SliceParams))),
                ("meta", C.POINTER(CtuMeta)), ("coeffs", C.POINTER(Coeffs))]


class FilterJob(C.Structure):
    _fields_ = [("pic", C.c_int32), ("pp", C.POINTER(PicParams)), ("sao", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double * NUM_KERNELS), ("kernel_launches", C.c_uint64 * NUM_KERNELS),
                ("intra_partitions", C.c_uint64), ("inter_partitions", C.c_uint64), ("coded_tus", (C.c_uint64 * 3) * 4)]


# ------------------------------------------------------------------------------------------------ helpers
def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p).value


def make_seq(width, height, bd_luma, bd_chroma=None, log2_ctu=6, max_pictures=8, strong_intra_smoothing=1, range_ext_flags=0):
    s = SeqParams()
    s.width, s.height = width, height
    s.bit_depth_luma = bd_luma
    s.bit_depth_chroma = bd_luma if bd_chroma is None else bd_chroma
    s.chroma_format = 1
    s.log2_ctu_size = log2_ctu
    s.max_pictures = max_pictures
    s.strong_intra_smoothing = strong_intra_smoothing
    s.range_ext_flags = range_ext_flags
    return s


def num_ctus(seq):
    c = 1 << seq.log2_ctu_size
    return ((seq.width + c - 1) // c) * ((seq.height + c - 1) // c)


def parts_per_ctu(seq):
    return 1 << (2 * seq.log2_ctu_size - 4)


META_ARRAYS = [("depth", np.uint8), ("part_size", np.int8), ("pred_mode", np.int8), ("qp", np.int8), ("tr_idx", np.uint8),
               ("cbf_y", np.uint8), ("cbf_u", np.uint8), ("cbf_v", np.uint8), ("ts_y", np.uint8), ("ts_u", np.uint8),
               ("ts_v", np.uint8), ("mv0", np.int16), ("mv1", np.int16), ("ref_idx0", np.int8), ("ref_idx1", np.int8),
               ("intra_dir_l", np.uint8), ("intra_dir_c", np.uint8), ("bypass", np.uint8), ("ipcm", np.uint8),
               ("slice_idx", np.uint16), ("tile_idx", np.uint16), ("ccp_u", np.int8), ("ccp_v", np.int8)]


class MetaHolder:
    """Keeps the numpy arrays alive next to the CtuMeta struct that points into them."""

    def __init__(self, arrays):
        self.arrays = {}
        for name, dt in META_ARRAYS:
            a = arrays.get(name)
            if a is not None:
                a = np.ascontiguousarray(a, dtype=dt)
            self.arrays[name] = a
        g = self.arrays
        m = CtuMeta()
        m.depth, m.part_size, m.pred_mode, m.qp, m.tr_idx = (_ptr(g[k]) for k in ("depth", "part_size", "pred_mode", "qp", "tr_idx"))
        for i, k in enumerate(("cbf_y", "cbf_u", "cbf_v")):
            m.cbf[i] = _ptr(g[k])
        for i, k in enumerate(("ts_y", "ts_u", "ts_v")):
            m.transform_skip[i] = _ptr(g[k])
        m.mv[0], m.mv[1] = _ptr(g["mv0"]), _ptr(g["mv1"])
        m.ref_idx[0], m.ref_idx[1] = _ptr(g["ref_idx0"]), _ptr(g["ref_idx1"])
        m.intra_dir[0], m.intra_dir[1] = _ptr(g["intra_dir_l"]), _ptr(g["intra_dir_c"])
        m.transquant_bypass, m.ipcm = _ptr(g["bypass"]), _ptr(g["ipcm"])
        m.slice_idx, m.tile_idx = _ptr(g["slice_idx"]), _ptr(g["tile_idx"])
        m.ccp_alpha[0], m.ccp_alpha[1] = _ptr(g["ccp_u"]), _ptr(g["ccp_v"])
        self.struct = m


class StagingHolder:
    """numpy views of a staging block (hmgpu_staging_alloc): .arrays[name] / .levels[k] / .starts[k] alias the page-locked memory;
    .struct / .coeffs are the structs to hand to the whole-picture calls.  fill() copies ordinary arrays in, levels in HM's dense
    layout; fill_compact() packs the levels (hmgpu_pack_levels: coded TUs only)."""

    def __init__(self, handle, meta_struct, coeff_struct, num_ctus, parts, ctu):
        self.handle, self.struct = handle, meta_struct
        self.num_ctus = num_ctus
        np_ = num_ctus * parts

        def view(addr, dt, n):
            return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt)
        m = meta_struct
        addr = {"depth": m.depth, "part_size": m.part_size, "pred_mode": m.pred_mode, "qp": m.qp, "tr_idx": m.tr_idx,
                "cbf_y": m.cbf[0], "cbf_u": m.cbf[1], "cbf_v": m.cbf[2], "ts_y": m.transform_skip[0], "ts_u": m.transform_skip[1],
                "ts_v": m.transform_skip[2], "mv0": m.mv[0], "mv1": m.mv[1], "ref_idx0": m.ref_idx[0], "ref_idx1": m.ref_idx[1],
                "intra_dir_l": m.intra_dir[0], "intra_dir_c": m.intra_dir[1], "bypass": m.transquant_bypass, "ipcm": m.ipcm,
                "slice_idx": m.slice_idx, "tile_idx": m.tile_idx}
        self.arrays = {}
        for name, dt in META_ARRAYS:
            if name not in addr:
                continue                                    # (arrays a staging block does not hold: cross-component prediction weights)
            n = num_ctus if name in ("slice_idx", "tile_idx") else (2 * np_ if name in ("mv0", "mv1") else np_)
            self.arrays[name] = view(addr[name], dt, n)
        self.levels = [view(coeff_struct.level[k], np.int16, num_ctus * ctu * ctu >> (2 if k else 0)) for k in range(3)]
        self.starts = [view(coeff_struct.ctu_level_start[k], np.uint32, num_ctus + 1) for k in range(3)]
        self._all = {"intra": (m.intra_dir[0], m.intra_dir[1]), "ts": tuple(m.transform_skip[k] for k in range(3)),
                     "bypass": m.transquant_bypass, "ipcm": m.ipcm}
        self._compact = coeff_struct
        self._dense = Coeffs()
        for k in range(3):
            self._dense.level[k] = coeff_struct.level[k]
        self.coeffs = self._dense

    def _fill_meta(self, meta):
        for name, _ in META_ARRAYS:
            if name not in self.arrays:
                continue
            src = meta.arrays.get(name)
            if src is not None:
                self.arrays[name][:] = src.reshape(-1)
            elif name in ("ref_idx0", "ref_idx1"):
                self.arrays[name][:] = -1
            else:
                self.arrays[name][:] = 0

    def set_groups(self, intra=True, flags=True):
        """leave optional groups of the block out of the copy: the intra modes (a picture without intra CUs), the transform-skip /
        lossless / PCM flags (a picture that uses none of them)"""
        m = self.struct
        m.intra_dir[0], m.intra_dir[1] = self._all["intra"] if intra else (None, None)
        for k in range(3):
            m.transform_skip[k] = self._all["ts"][k] if flags else None
        m.transquant_bypass = self._all["bypass"] if flags else None
        m.ipcm = self._all["ipcm"] if flags else None

    def fill(self, meta, coeffs):
        self._fill_meta(meta)
        for k in range(3):
            self.levels[k][:] = coeffs.arrays[k].reshape(-1)
        self.coeffs = self._dense

    def fill_compact(self, lib, seq, meta, coeffs):
        """lib: the loaded libhmgpu (hmgpu_pack_levels is host code); returns the number of level bytes that will cross the bus"""
        self._fill_meta(meta)
        lv = (C.c_void_p * 3)(*[self._compact.level[k] for k in range(3)])
        stt = (C.c_void_p * 3)(*[self._compact.ctu_level_start[k] for k in range(3)])
        st = lib.hmgpu_pack_levels(C.byref(seq), C.byref(meta.struct), C.byref(coeffs.struct), lv, stt)
        if st != 0:
            raise RuntimeError("hmgpu_pack_levels: status %d" % st)
        self.coeffs = self._compact
        return 2 * int(sum(int(self.starts[k][self.num_ctus]) for k in range(3)))


class CoeffHolder:
    def __init__(self, y, cb, cr, pcm=None):
        self.arrays = [np.ascontiguousarray(a, dtype=np.int16) for a in (y, cb, cr)]
        self.pcm = [np.ascontiguousarray(a, dtype=np.int16) for a in pcm] if pcm is not None else None
        self.struct = Coeffs()
        for i in range(3):
            self.struct.level[i] = _ptr(self.arrays[i])
            if self.pcm is not None:
                self.struct.pcm_sample[i] = _ptr(self.pcm[i])


def make_slice(slice_type, ref_pic=((), ()), ref_poc=((), ()), cb_qp_offset=0, cr_qp_offset=0, pps_cb=0, pps_cr=0,
               deblocking_disable=0, beta_offset_div2=0, tc_offset_div2=0, lf_across_slices=1, lf_across_tiles=1):
    s = SliceParams()
    s.slice_type = slice_type
    s.cb_qp_offset, s.cr_qp_offset = cb_qp_offset, cr_qp_offset
    s.pps_cb_qp_offset, s.pps_cr_qp_offset = pps_cb, pps_cr
    s.deblocking_disable = deblocking_disable
    s.beta_offset_div2, s.tc_offset_div2 = beta_offset_div2, tc_offset_div2
    s.lf_across_slices = lf_across_slices
    s.lf_across_tiles = lf_across_tiles
    for l in range(2):
        s.num_ref_idx[l] = len(ref_pic[l])
        for i in range(MAX_REF):
            s.ref_pic[l][i] = NO_PIC
        for i, (h, p) in enumerate(zip(ref_pic[l], ref_poc[l])):
            s.ref_pic[l][i] = int(h)
            s.ref_poc[l][i] = int(p)
    return s


def clone_slice(s):
    """byte copy of a SliceParams (copy.copy refuses ctypes structures that hold pointers)"""
    return SliceParams.from_buffer_copy(s)


def make_pic_params(sao_enabled=1, lf_across_tiles=1, sao_offset_shift=(0, 0)):
    p = PicParams()
    p.sao_enabled, p.lf_across_tiles = sao_enabled, lf_across_tiles
    p.sao_offset_shift_luma, p.sao_offset_shift_chroma = sao_offset_shift
    return p


def sao_array_from_raw(raw):
    """raw: int32 [num_ctus, 3, 35] (mode, type, aux, offset[32]) -> ctypes array of SaoParam [num_ctus*3]"""
    raw = np.ascontiguousarray(raw, dtype=np.int32)
    n = raw.shape[0] * 3
    arr = (SaoParam * n)()
    C.memmove(arr, raw.ctypes.data, n * C.sizeof(SaoParam))
    return arr


def sao_array_to_np(arr, n_ctus):
    out = np.zeros((n_ctus, 3, 35), dtype=np.int32)
    C.memmove(out.ctypes.data, arr, n_ctus * 3 * C.sizeof(SaoParam))
    return out
