"""ctypes binding of oracle/libhmoracle.so (our plain-C restatement of the HM pixel path, oracle/hm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
checker / reported CPU baseline.  The product (libhm_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from libhm_amd import abi

class OraclePicture(C.Structure):           # hm_oracle.h: hmo_picture
    _fields_ = [("plane", C.c_void_p * 3)]


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhmoracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libhmoracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.hmo_decompress_ctus.restype = C.c_int
        _lib.hmo_loop_filter_pic.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pic(planes):
    p = OraclePicture()
    for i in range(3):
        p.plane[i] = planes[i].ctypes.data_as(C.c_void_p).value
    return p


# ------------------------------------------------------------------------------------------------ kernel level
def itr(bit_depth, coeff, use_dst):
    coeff = np.ascontiguousarray(coeff, dtype=np.int32)
    n_tu, n, _ = coeff.shape
    out = np.empty_like(coeff)
    for i in range(n_tu):
        lib().hmo_itr(bit_depth, _p(coeff[i]), _p(out[i]), n, int(use_dst))
    return out


def inverse_transform_tus(levels, log2_size, bit_depth, qp_per, qp_rem, flags):
    """levels int16 [n, N, N] -> resid int16 [n, N, N]"""
    levels = np.ascontiguousarray(levels, dtype=np.int16)
    n_tu = levels.shape[0]
    size = 1 << log2_size
    out = np.zeros((n_tu, size, size), dtype=np.int16)
    f = lib().hmo_inverse_transform_tu
    for i in range(n_tu):
        f(_p(levels[i]), _p(out[i]), size, log2_size, bit_depth, int(qp_per[i]), int(qp_rem[i]), int(flags[i]))
    return out


def qp_param(qp_y, comp, bit_depth, chroma_qp_offset=0):
    per, rem = C.c_int(), C.c_int()
    lib().hmo_qp_param(int(qp_y), comp, bit_depth, chroma_qp_offset, C.byref(per), C.byref(rem))
    return per.value, rem.value


def pred_inter_blk(is_chroma, bit_depth, plane, bx, by, w, h, mvx, mvy, bi):
    plane = np.ascontiguousarray(plane, dtype=np.int16)
    dst = np.zeros((h, w), dtype=np.int16)
    lib().hmo_pred_inter_blk(int(is_chroma), bit_depth, _p(plane), plane.shape[1], plane.shape[1], plane.shape[0],
                             bx, by, w, h, mvx, mvy, int(bi), _p(dst), w)
    return dst


def add_avg(a, b, bit_depth):
    a = np.ascontiguousarray(a, dtype=np.int16)
    b = np.ascontiguousarray(b, dtype=np.int16)
    dst = np.zeros_like(a)
    lib().hmo_add_avg(_p(a), _p(b), _p(dst), a.shape[1], a.shape[0], a.shape[1], bit_depth)
    return dst


def sao_offset_block(bit_depth, type_idx, offset32, plane, x0, y0, w, h, avail8):
    plane = np.ascontiguousarray(plane, dtype=np.int16)
    res = plane.copy()
    stride = plane.shape[1]
    off = np.ascontiguousarray(offset32, dtype=np.int32)
    av = np.ascontiguousarray(avail8, dtype=np.int32)
    o = 2 * (y0 * stride + x0)
    lib().hmo_sao_offset_block(bit_depth, type_idx, _p(off), C.c_void_p(plane.ctypes.data + o), C.c_void_p(res.ctypes.data + o),
                               stride, stride, w, h, _p(av))
    return res


# ------------------------------------------------------------------------------------------------ picture level
def decompress_ctus(seq, slices, meta, coeffs, cur_planes, ref_planes_list, first_ctu=0, num_ctus=None):
    """cur_planes: list of 3 int16 arrays (modified in place).  ref_planes_list: list indexed by picture handle."""
    n = abi.num_ctus(seq) if num_ctus is None else num_ctus
    cur = _pic(cur_planes)
    refs = (OraclePicture * max(1, len(ref_planes_list)))()
    for i, rp in enumerate(ref_planes_list):
        if rp is not None:
            refs[i] = _pic(rp)
    n_intra = C.c_int64(0)
    sl = (abi.SliceParams * len(slices))(*slices)
    r = lib().hmo_decompress_ctus(C.byref(seq), sl, C.byref(meta.struct), C.byref(coeffs.struct), C.byref(cur), refs,
                                  len(ref_planes_list), first_ctu, n, C.byref(n_intra))
    assert r == 0, r
    return n_intra.value


def plane_hashes(planes, bit_depths):
    """(crc bytes [6], checksum bytes [12]) of a picture, HM's decoded-picture-hash conventions"""
    crc, chk = np.zeros(6, dtype=np.uint8), np.zeros(12, dtype=np.uint8)
    for c, (pl, bd) in enumerate(zip(planes, bit_depths)):
        pl = np.ascontiguousarray(pl, dtype=np.int16)
        o2, o4 = (C.c_uint8 * 2)(), (C.c_uint8 * 4)()
        lib().hmo_plane_crc(bd, pl.ctypes.data_as(C.c_void_p), pl.shape[1], pl.shape[0], pl.shape[1], o2)
        lib().hmo_plane_checksum(bd, pl.ctypes.data_as(C.c_void_p), pl.shape[1], pl.shape[0], pl.shape[1], o4)
        crc[2 * c:2 * c + 2] = list(o2)
        chk[4 * c:4 * c + 4] = list(o4)
    return crc, chk


def loop_filter_pic(seq, slices, meta, pp, planes, dir_mask=3):
    pic = _pic(planes)
    sl = (abi.SliceParams * len(slices))(*slices)
    r = lib().hmo_loop_filter_pic(C.byref(seq), sl, C.byref(meta.struct), C.byref(pp), C.byref(pic), dir_mask)
    assert r == 0, r


def boundary_strengths(seq, slices, meta, pp):
    n = abi.num_ctus(seq) * abi.parts_per_ctu(seq)
    bv = np.zeros(n, dtype=np.uint8)
    bh = np.zeros(n, dtype=np.uint8)
    sl = (abi.SliceParams * len(slices))(*slices)
    r = lib().hmo_boundary_strengths(C.byref(seq), sl, C.byref(meta.struct), C.byref(pp), _p(bv), _p(bh))
    assert r == 0, r
    return bv.reshape(abi.num_ctus(seq), -1), bh.reshape(abi.num_ctus(seq), -1)


def sao_reconstruct_params(seq, pp, meta, raw):
    n = abi.num_ctus(seq)
    raw_arr = abi.sao_array_from_raw(raw)
    rec_arr = (abi.SaoParam * (n * 3))()
    r = lib().hmo_sao_reconstruct_params(C.byref(seq), C.byref(pp), C.byref(meta.struct), raw_arr, rec_arr)
    assert r == 0, r
    return abi.sao_array_to_np(rec_arr, n)


def sao_process(seq, slices, pp, meta, rec, src_planes):
    dst = [p.copy() for p in src_planes]
    rec_arr = abi.sao_array_from_raw(rec)
    sl = (abi.SliceParams * len(slices))(*slices)
    s, d = _pic(src_planes), _pic(dst)
    r = lib().hmo_sao_process(C.byref(seq), sl, C.byref(pp), C.byref(meta.struct), rec_arr, C.byref(s), C.byref(d))
    assert r == 0, r
    return dst
