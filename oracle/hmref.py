"""ctypes binding of oracle/_ref/libhmref.so (the REAL HM 16.0 libraries + oracle/ref_harness.cpp).

TEST INFRASTRUCTURE ONLY: used by oracle/make_golden.py (fixture generation, builder container) and,
when the library is present, by tests/ and bench.py's cpu_baseline leg.  The product never imports it.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libhmref.so")
ENCODER_PATH = os.path.join(_HERE, "_ref", "TAppEncoder")


def available():
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_dec_open.restype = C.c_void_p
        _lib.ref_dec_open.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        for n in ("ref_dec_close", "ref_dec_info", "ref_dec_slices", "ref_dec_sao_params", "ref_dec_wp", "ref_dec_scaling_lists", "ref_dec_tile_idx", "ref_dec_pcm", "ref_dec_pcm_info", "ref_dec_hashes"):
            getattr(_lib, n).restype = None
        _lib.ref_dec_next.argtypes = [C.c_void_p]
        _lib.ref_dec_filter_step.argtypes = [C.c_void_p]
        _lib.ref_dec_finish.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_close.argtypes = [C.c_void_p]
        _lib.ref_dec_info.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_slices.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_sao_params.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_wp.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_scaling_lists.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_tile_idx.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_pcm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.ref_dec_pcm_info.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_hashes.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dec_coeffs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.ref_dec_ccp_alpha.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.ref_dec_ccp_alpha.restype = None
        _lib.ref_dec_planes.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        _lib.ref_dec_dpb_planes.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3
        _lib.ref_dec_meta.argtypes = [C.c_void_p] + [C.c_void_p] * 22
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------------------- KATs
def kat_init(bdY, bdC):
    lib().ref_kat_init(bdY, bdC)


def kat_itr(bit_depth, coeff, use_dst):
    """coeff: int32 [n, h, w] -> residual int32 [n, h, w] through HM xITrMxN"""
    coeff = np.ascontiguousarray(coeff, dtype=np.int32)
    n, h, w = coeff.shape
    out = np.empty_like(coeff)
    lib().ref_kat_itr(C.c_int(bit_depth), _p(coeff), _p(out), w, h, int(use_dst), n)
    return out


def kat_interp(comp, plane, x0, y0, w, h, xfrac, yfrac, bi):
    """plane int16 2-D (needs 4 samples of margin around the block)"""
    plane = np.ascontiguousarray(plane, dtype=np.int16)
    stride = plane.shape[1]
    dst = np.zeros((h, w), dtype=np.int16)
    src_ptr = plane.ctypes.data + 2 * (y0 * stride + x0)
    lib().ref_kat_interp(comp, C.c_void_p(src_ptr), stride, _p(dst), w, w, h, xfrac, yfrac, int(bi))
    return dst


def kat_addavg(s0, s1):
    s0 = np.ascontiguousarray(s0, dtype=np.int16)
    s1 = np.ascontiguousarray(s1, dtype=np.int16)
    h, w = s0.shape
    dst = np.zeros((h, w), dtype=np.int16)
    lib().ref_kat_addavg(_p(s0), _p(s1), _p(dst), w, h)
    return dst


def kat_sao_block(comp, bdY, bdC, type_idx, offset32, plane, x0, y0, w, h, avail8):
    """plane int16 2-D with >= 1 sample margin around block; returns copy of plane with block filtered"""
    plane = np.ascontiguousarray(plane, dtype=np.int16)
    res = plane.copy()
    stride = plane.shape[1]
    off = np.ascontiguousarray(offset32, dtype=np.int32)
    av = np.ascontiguousarray(avail8, dtype=np.int32)
    o = 2 * (y0 * stride + x0)
    lib().ref_kat_sao_block(comp, bdY, bdC, type_idx, _p(off), C.c_void_p(plane.ctypes.data + o),
                            C.c_void_p(res.ctypes.data + o), stride, stride, w, h, _p(av))
    return res


# ----------------------------------------------------------------------------------------- decoder
META_FIELDS = [("depth", np.uint8, 1), ("part_size", np.int8, 1), ("pred_mode", np.int8, 1), ("qp", np.int8, 1),
               ("tr_idx", np.uint8, 1), ("cbf_y", np.uint8, 1), ("cbf_u", np.uint8, 1), ("cbf_v", np.uint8, 1),
               ("ts_y", np.uint8, 1), ("ts_u", np.uint8, 1), ("ts_v", np.uint8, 1),
               ("mv0", np.int16, 2), ("mv1", np.int16, 2), ("ref_idx0", np.int8, 1), ("ref_idx1", np.int8, 1),
               ("intra_dir_l", np.uint8, 1), ("intra_dir_c", np.uint8, 1), ("bypass", np.uint8, 1),
               ("ipcm", np.uint8, 1), ("skip", np.uint8, 1), ("merge", np.uint8, 1)]


def chroma_scale(info):
    """(sx, sy) of the chroma planes (getComponentScaleX / Y); 4:0:0 streams keep 4:2:0-shaped dummies"""
    f = info["chroma_format"]
    return (0 if f == 3 else 1, 1 if f in (0, 1) else 0)


def chroma_shift(info):
    return sum(chroma_scale(info))


class RefDecoder:
    """Iterate over the pictures of an Annex-B stream in decode order, exposing HM's state around the filter stage."""

    def __init__(self, bitstream: bytes, check_hash=True):
        self._buf = np.frombuffer(bitstream, dtype=np.uint8).copy()
        self._h = lib().ref_dec_open(_p(self._buf), len(self._buf), int(check_hash))

    def close(self):
        if self._h:
            lib().ref_dec_close(self._h)
            self._h = None

    def next(self):
        return bool(lib().ref_dec_next(self._h))

    def info(self):
        a = np.zeros(16, dtype=np.int32)
        lib().ref_dec_info(self._h, _p(a))
        keys = ["width", "height", "bd_y", "bd_c", "poc", "slice_type", "num_ctus", "ctus_w", "parts", "ctu_size",
                "num_slices", "use_sao", "lf_across_tiles", "chroma_format", "tid", "max_depth"]
        return dict(zip(keys, (int(v) for v in a)))

    def slices(self, n):
        a = np.zeros((n, 64), dtype=np.int32)
        lib().ref_dec_slices(self._h, _p(a))
        return a

    def pcm(self, info):
        """(info[5], [3 sample arrays])"""
        pi = np.zeros(5, dtype=np.int32)
        lib().ref_dec_pcm_info(self._h, _p(pi))
        n, cs = info["num_ctus"], info["ctu_size"]
        res = []
        for c in range(3):
            a = np.zeros((n, (cs * cs) >> (chroma_shift(info) if c else 0)), dtype=np.int16)
            lib().ref_dec_pcm(self._h, c, _p(a))
            res.append(a)
        return pi, res

    def tile_idx(self, num_ctus):
        a = np.zeros(num_ctus, dtype=np.int32)
        lib().ref_dec_tile_idx(self._h, _p(a))
        return a

    def scaling_lists(self):
        a = np.zeros(1 + 4 * 6 * 64 + 4 * 6, dtype=np.int32)
        lib().ref_dec_scaling_lists(self._h, _p(a))
        return a

    def wp(self, n):
        a = np.zeros((n, 195), dtype=np.int32)
        lib().ref_dec_wp(self._h, _p(a))
        return a

    def meta(self, info):
        n, parts = info["num_ctus"], info["parts"]
        out = {}
        ptrs = []
        for name, dt, k in META_FIELDS:
            shape = (n, parts) if k == 1 else (n, parts, k)
            out[name] = np.zeros(shape, dtype=dt)
            ptrs.append(_p(out[name]))
        out["slice_idx"] = np.zeros(n, dtype=np.int32)
        ptrs.append(_p(out["slice_idx"]))
        lib().ref_dec_meta(self._h, *ptrs)
        return out

    def ccp_alpha(self, info):
        """cross-component prediction weights [2][num_ctus, parts] (Cb, Cr)"""
        res = []
        for c in (1, 2):
            a = np.zeros((info["num_ctus"], info["parts"]), dtype=np.int8)
            lib().ref_dec_ccp_alpha(self._h, c, _p(a))
            res.append(a)
        return res

    def coeffs(self, info):
        n, cs = info["num_ctus"], info["ctu_size"]
        res = []
        for c in range(3):
            a = np.zeros((n, (cs * cs) >> (chroma_shift(info) if c else 0)), dtype=np.int32)
            lib().ref_dec_coeffs(self._h, c, _p(a))
            res.append(a)
        return res

    def sao_params(self, info):
        a = np.zeros((info["num_ctus"], 3, 35), dtype=np.int32)
        lib().ref_dec_sao_params(self._h, _p(a))
        return a

    def planes(self, info):
        w, h = info["width"], info["height"]
        sx, sy = chroma_scale(info)
        y = np.zeros((h, w), dtype=np.int16)
        cb = np.zeros((h >> sy, w >> sx), dtype=np.int16)
        cr = np.zeros((h >> sy, w >> sx), dtype=np.int16)
        lib().ref_dec_planes(self._h, _p(y), _p(cb), _p(cr))
        return y, cb, cr

    def filter_step(self):
        return int(lib().ref_dec_filter_step(self._h))

    def hashes(self):
        a = np.zeros(18, dtype=np.uint8)
        lib().ref_dec_hashes(self._h, _p(a))
        return a[:6].copy(), a[6:].copy()

    def finish(self):
        md5 = np.zeros(48, dtype=np.uint8)
        ok = lib().ref_dec_finish(self._h, _p(md5))
        return bool(ok), md5
