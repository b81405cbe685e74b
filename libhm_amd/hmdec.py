"""ctypes mirror of include/hmdec.h (libhmdec.so): the libHMDecoder-compatible decoder on top of the device path.
Used by the tests and tools; applications written against libHM's libHMDecoder.h link the .so directly."""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

LIBHMDEC_OK = 0


def lib():
    global _lib
    if _lib is None:
        from . import lib as gpu_lib          # libhmgpu first (shares the HIP runtime with torch when torch is loaded)
        gpu_lib()
        path = os.path.join(HERE, "libhmdec.so")
        if not os.path.exists(path):
            raise RuntimeError("libhmdec.so is missing: run `python libhm_amd/build.py`")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.libHMDec_get_version.restype = C.c_char_p
        L.libHMDec_new_decoder.restype = C.c_void_p
        L.libHMDec_free_decoder.argtypes = [C.c_void_p]
        L.libHMDec_set_SEI_Check.argtypes = [C.c_void_p, C.c_bool]
        L.libHMDec_set_max_temporal_layer.argtypes = [C.c_void_p, C.c_int]
        L.libHMDec_push_nal_unit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_bool, C.POINTER(C.c_bool), C.POINTER(C.c_bool)]
        L.libHMDec_get_picture.argtypes = [C.c_void_p]
        L.libHMDec_get_picture.restype = C.c_void_p
        for f in ("libHMDEC_get_picture_width", "libHMDEC_get_picture_height", "libHMDEC_get_picture_stride"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.libHMDEC_get_POC.argtypes = [C.c_void_p]
        L.libHMDEC_get_image_plane.argtypes = [C.c_void_p, C.c_int]
        L.libHMDEC_get_image_plane.restype = C.POINTER(C.c_int16)
        L.libHMDEC_get_chroma_format.argtypes = [C.c_void_p]
        L.libHMDEC_get_internal_bit_depth.argtypes = [C.c_int]
        L.hmdec_set_device.argtypes = [C.c_void_p, C.c_int]
        L.hmdec_set_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.hmdec_num_devices.argtypes = [C.c_void_p]
        L.hmdec_transfer_bytes.argtypes = [C.c_void_p]
        L.hmdec_transfer_bytes.restype = C.c_ulonglong
        L.hmdec_set_parse_only.argtypes = [C.c_void_p, C.c_int]
        L.hmdec_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.hmdec_hash_mismatches.argtypes = [C.c_void_p]
        L.hmdec_pictures_decoded.argtypes = [C.c_void_p]
        L.hmdec_device_batches.argtypes = [C.c_void_p]
        L.hmdec_picture_range_ext_flags.argtypes = [C.c_void_p]
        L.hmdec_picture_chroma_format.argtypes = [C.c_void_p]
        L.hmdec_picture_sao_offset_shift.argtypes = [C.c_void_p, C.c_int]
        L.hmdec_set_device_md5.argtypes = [C.c_void_p, C.c_int]
        L.hmdec_last_error.argtypes = [C.c_void_p]
        L.hmdec_last_error.restype = C.c_char_p
        L.hmdec_last_decoded_picture.argtypes = [C.c_void_p]
        L.hmdec_last_decoded_picture.restype = C.c_void_p
        L.hmdec_open_picture.argtypes = [C.c_void_p]
        L.hmdec_open_picture.restype = C.c_void_p
        L.hmdec_picture_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.hmdec_picture_num_slices.argtypes = [C.c_void_p]
        L.hmdec_picture_slice_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.hmdec_picture_hash_sei.argtypes = [C.c_void_p, C.c_void_p]
        L.hmdec_picture_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.hmdec_picture_conformance_window.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.hmdec_internal_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.POINTER(BlockValue))]
        _lib = L
    return _lib


class BlockValue(C.Structure):
    _fields_ = [("x", C.c_ushort), ("y", C.c_ushort), ("w", C.c_ushort), ("h", C.c_ushort), ("value", C.c_int), ("value2", C.c_int)]


INFO = {n: i for i, n in enumerate(
    ["CTU_SLICE_INDEX", "CU_PREDICTION_MODE", "CU_TRQ_BYPASS", "CU_SKIP_FLAG", "CU_PART_MODE", "CU_INTRA_MODE_LUMA", "CU_INTRA_MODE_CHROMA",
     "CU_ROOT_CBF", "PU_MERGE_FLAG", "PU_MERGE_INDEX", "PU_UNI_BI_PREDICTION", "PU_REFERENCE_POC_0", "PU_MV_0", "PU_REFERENCE_POC_1", "PU_MV_1",
     "TU_CBF_Y", "TU_CBF_CB", "TU_CBF_CR", "TU_COEFF_TR_SKIP_Y", "TU_COEFF_TR_SKIP_Cb", "TU_COEFF_TR_SKIP_Cr", "TU_COEFF_ENERGY_Y",
     "TU_COEFF_ENERGY_CB", "TU_COEFF_ENERGY_CR"])}


def split_nal_units(stream):
    """Annex B byte stream -> list of NAL units (without start codes)"""
    b = bytes(stream)
    starts, i = [], 0
    while True:
        j = b.find(b"\x00\x00\x01", i)
        if j < 0:
            break
        starts.append(j + 3)
        i = j + 3
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else len(b)
        while e > s and b[e - 1] == 0:           # trailing_zero_8bits / the zero_byte of the next start code
            e -= 1
        out.append(b[s:e])
    return out


_DTYPES = {"depth": np.uint8, "part_size": np.int8, "pred_mode": np.int8, "qp": np.int8, "tr_idx": np.uint8, "bypass": np.uint8,
           "ipcm": np.uint8, "skip": np.uint8, "merge": np.uint8, "slice_idx": np.uint16, "tile_idx": np.uint16, "sao": np.int32}


class Picture:
    """a decoded picture handle (valid until the decoder reuses the buffer)"""

    def __init__(self, handle):
        self.h = handle

    @property
    def poc(self):
        return lib().libHMDEC_get_POC(self.h)

    def size(self, c=0):
        return lib().libHMDEC_get_picture_width(self.h, c), lib().libHMDEC_get_picture_height(self.h, c)

    def plane(self, c):
        w, h = self.size(c)
        stride = lib().libHMDEC_get_picture_stride(self.h, c)
        p = lib().libHMDEC_get_image_plane(self.h, c)
        if not p:
            raise RuntimeError("no samples: the decoder runs parse-only or the picture was never reconstructed")
        return np.ctypeslib.as_array(p, shape=(h, stride))[:, :w].copy()

    def array(self, name):
        ptr, n = C.c_void_p(), C.c_int64()
        if lib().hmdec_picture_array(self.h, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        base = name.rstrip("012")
        dt = _DTYPES.get(name, _DTYPES.get(base, None))
        if dt is None:
            dt = {"cbf": np.uint8, "ts": np.uint8, "mv": np.int16, "ref_idx": np.int8, "intra_dir": np.uint8, "coeff": np.int16,
                  "pcm": np.int16, "plane": np.int16, "ccp": np.int8}[base]
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def geometry(self):
        g = (C.c_int32 * 12)()
        lib().hmdec_picture_geometry(self.h, g)
        keys = ("width", "height", "log2_ctb", "bd_y", "bd_c", "pcm_bd_y", "pcm_bd_c", "pcm_lf_disable", "strong_intra", "sao", "lf_across_tiles", "num_ctbs")
        out = dict(zip(keys, (int(v) for v in g)))
        out["range_ext"] = int(lib().hmdec_picture_range_ext_flags(self.h))
        out["chroma_format"] = int(lib().hmdec_picture_chroma_format(self.h))
        out["csx"], out["csy"] = (0 if out["chroma_format"] == 3 else 1), (1 if out["chroma_format"] in (0, 1) else 0)
        out["sao_shift"] = (int(lib().hmdec_picture_sao_offset_shift(self.h, 0)), int(lib().hmdec_picture_sao_offset_shift(self.h, 1)))
        return out

    def conformance_window(self):
        w = (C.c_int32 * 4)()
        lib().hmdec_picture_conformance_window(self.h, w)
        return tuple(w)

    def cropped_plane(self, c):
        g = self.geometry()
        l, r, t, b = self.conformance_window()
        if c:
            l, r, t, b = l >> g["csx"], r >> g["csx"], t >> g["csy"], b >> g["csy"]
        p = self.plane(c)
        return p[t:p.shape[0] - b, l:p.shape[1] - r]

    def num_slices(self):
        return lib().hmdec_picture_num_slices(self.h)

    def slice_params(self, i):
        sp, sl = abi.SliceParams(), abi.ScalingLists()
        if lib().hmdec_picture_slice_params(self.h, i, C.byref(sp), C.byref(sl)) != 0:
            raise IndexError(i)
        return sp, sl

    def hash_sei(self):
        d = (C.c_uint8 * 48)()
        m = lib().hmdec_picture_hash_sei(self.h, d)
        return m, bytes(d)


class Decoder:
    def __init__(self, parse_only=False, device=0, check_hash=True, max_temporal_layer=-1, threads=1, device_md5=None, devices=None):
        """devices: GPU ordinals of several device contexts (hmdec_set_devices; the same ordinal twice = two contexts on one GPU)"""
        self.ctx = lib().libHMDec_new_decoder()
        if not self.ctx:
            raise MemoryError("libHMDec_new_decoder")
        lib().hmdec_set_parse_only(self.ctx, 1 if parse_only else 0)
        lib().hmdec_set_device(self.ctx, device)
        if devices:
            lib().hmdec_set_devices(self.ctx, (C.c_int * len(devices))(*devices), len(devices))
        lib().hmdec_set_threads(self.ctx, threads)
        lib().libHMDec_set_SEI_Check(self.ctx, check_hash)
        lib().libHMDec_set_max_temporal_layer(self.ctx, max_temporal_layer)
        if device_md5 is not None:                 # MD5 hash SEIs checked on the device (default: the decoder's hash threads / HMDEC_DEVICE_MD5)
            lib().hmdec_set_device_md5(self.ctx, 1 if device_md5 else 0)

    def close(self):
        if self.ctx:
            lib().libHMDec_free_decoder(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def push(self, nal, eof=False):
        """returns (new_picture, check_output)"""
        new_pic, check = C.c_bool(False), C.c_bool(False)
        buf = (C.c_uint8 * len(nal)).from_buffer_copy(nal)
        r = lib().libHMDec_push_nal_unit(self.ctx, buf, len(nal), eof, C.byref(new_pic), C.byref(check))
        if r != LIBHMDEC_OK:
            raise RuntimeError("libHMDec_push_nal_unit failed (%d): %s" % (r, lib().hmdec_last_error(self.ctx).decode()))
        return new_pic.value, check.value

    def get_picture(self):
        h = lib().libHMDec_get_picture(self.ctx)
        return Picture(h) if h else None

    def internal_info(self, pic, kind):
        """libHMDEC_get_internal_info as a list of (x, y, w, h, value, value2)"""
        data = C.POINTER(BlockValue)()
        n = lib().hmdec_internal_info(self.ctx, pic.h, INFO[kind], C.byref(data))
        if n < 0:
            raise RuntimeError("hmdec_internal_info")
        return [(data[i].x, data[i].y, data[i].w, data[i].h, data[i].value, data[i].value2) for i in range(n)]

    def last_decoded(self):
        h = lib().hmdec_last_decoded_picture(self.ctx)
        return Picture(h) if h else None

    @property
    def hash_mismatches(self):
        return lib().hmdec_hash_mismatches(self.ctx)

    @property
    def pictures_decoded(self):
        return lib().hmdec_pictures_decoded(self.ctx)

    @property
    def device_batches(self):
        return lib().hmdec_device_batches(self.ctx)

    @property
    def num_devices(self):
        return lib().hmdec_num_devices(self.ctx)

    @property
    def transfer_bytes(self):
        return int(lib().hmdec_transfer_bytes(self.ctx))

    def decode_stream(self, stream, on_decoded=None, on_output=None):
        """libHM's documented loop (libHMDecoder.h:36-77) over an Annex B stream"""
        nals = split_nal_units(stream)
        seen = 0
        for i, nal in enumerate(nals):
            eof = i == len(nals) - 1
            while True:
                new_pic, check = self.push(nal, eof)
                if on_decoded and self.pictures_decoded > seen:
                    seen = self.pictures_decoded
                    on_decoded(self.last_decoded())
                if check:
                    while True:
                        p = self.get_picture()
                        if p is None:
                            break
                        if on_output:
                            on_output(p)
                if not new_pic:
                    break
