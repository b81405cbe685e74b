"""libhm_amd -- MI355X-native pixel reconstruction for the HM (HEVC) decoder.

The product is libhm_amd/libhmgpu.so: hand-written HIP kernels for gfx950 behind the C ABI of include/hmgpu.h
(drop-in for HM's TDecGop::decompressSlice / filterPicture).  This package is only the thin ctypes
binding used by tests and bench.py; it contains no compute and NO fallback: if the shared library or a GPU is missing,
every entry point raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhmgpu.so")
_lib = None


class HmgpuError(RuntimeError):
    def __init__(self, status, what, device_error=0):
        self.status = status
        self.device_error = device_error
        names = {1: "HMGPU_EINVAL", 2: "HMGPU_EDEVICE", 3: "HMGPU_EUNSUPPORTED", 4: "HMGPU_ENOMEM"}
        super().__init__("%s failed: %s%s" % (what, names.get(status, status),
                                              " (hipError %d)" % device_error if status == 2 else ""))


def _share_torch_hip_runtime():
    """One HIP runtime per process: PyTorch ships its own libamdhip64 and a process that has already initialised the
    system copy (through libhmgpu.so) can no longer bring up torch's ("No HIP GPUs are available").  The frame-parallel
    path needs both (RCCL through torch.distributed on this library's stream), so when torch is installed its runtime is
    loaded first and libhmgpu.so's NEEDED libamdhip64 resolves to it (same soname)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        path = os.path.join(os.path.dirname(spec.origin), "lib", name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def lib():
    """Load libhmgpu.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libhm_amd/libhmgpu.so is missing: build it with `python libhm_amd/build.py` "
                               "(or __graft_entry__.build()); there is no CPU fallback")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.hmgpu_create.argtypes = [C.POINTER(abi.SeqParams), C.c_int, C.POINTER(C.c_void_p)]
        L.hmgpu_destroy.argtypes = [C.c_void_p]
        L.hmgpu_destroy.restype = None
        L.hmgpu_last_device_error.argtypes = [C.c_void_p]
        L.hmgpu_debug_stall_intra.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.hmgpu_sync.argtypes = [C.c_void_p]
        L.hmgpu_status_string.restype = C.c_char_p
        L.hmgpu_kernel_name.restype = C.c_char_p
        L.hmgpu_num_ctus.argtypes = [C.POINTER(abi.SeqParams)]
        L.hmgpu_parts_per_ctu.argtypes = [C.POINTER(abi.SeqParams)]
        L.hmgpu_picture_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.hmgpu_picture_release.argtypes = [C.c_void_p, C.c_int32]
        L.hmgpu_picture_upload.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.hmgpu_picture_download.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.hmgpu_set_streams.argtypes = [C.c_void_p, C.c_int32]
        L.hmgpu_picture_download_packed.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)] + [C.c_int32] * 5
        L.hmgpu_picture_hash.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
        L.hmgpu_picture_device_region.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.hmgpu_picture_commit_received.argtypes = [C.c_void_p, C.c_int32]
        L.hmgpu_picture_transfer.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.hmgpu_transfer_bytes.argtypes = [C.c_void_p]
        L.hmgpu_transfer_bytes.restype = C.c_uint64
        L.hmgpu_stream.argtypes = [C.c_void_p]
        L.hmgpu_stream.restype = C.c_void_p
        L.hmgpu_decompress_slice.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(abi.SliceParams), C.POINTER(abi.CtuMeta),
                                             C.POINTER(abi.Coeffs), C.c_int32, C.c_int32]
        L.hmgpu_filter_picture.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.PicParams), C.c_void_p]
        L.hmgpu_filter_picture_stages.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.PicParams), C.c_void_p, C.c_int32]
        L.hmgpu_decompress_pictures.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.PictureJob)]
        L.hmgpu_filter_pictures.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.FilterJob)]
        L.hmgpu_staging_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(abi.CtuMeta), C.POINTER(abi.Coeffs)]
        L.hmgpu_staging_free.argtypes = [C.c_void_p, C.c_void_p]
        L.hmgpu_staging_free.restype = None
        L.hmgpu_pack_levels.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(abi.CtuMeta), C.POINTER(abi.Coeffs), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.hmgpu_replay.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.hmgpu_replay_batch.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32]
        L.hmgpu_set_profiling.argtypes = [C.c_void_p, C.c_int32]
        L.hmgpu_get_stats.argtypes = [C.c_void_p, C.POINTER(abi.Stats), C.c_int32]
        L.hmgpu_inverse_transform_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5
        L.hmgpu_mc_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_levels(seq, meta, coeffs):
    """HM's dense level arrays -> a CoeffHolder with compact levels (coded TUs only) + CTU starts (hmgpu_pack_levels: host code)"""
    n = abi.num_ctus(seq)
    ctu = 1 << seq.log2_ctu_size
    out = abi.CoeffHolder(*[np.zeros(n * ctu * ctu >> (2 if k else 0), dtype=np.int16) for k in range(3)])
    out.starts = [np.zeros(n + 1, dtype=np.uint32) for _ in range(3)]
    lv = (C.c_void_p * 3)(*[out.struct.level[k] for k in range(3)])
    stt = (C.c_void_p * 3)(*[s.ctypes.data for s in out.starts])
    st = lib().hmgpu_pack_levels(C.byref(seq), C.byref(meta.struct), C.byref(coeffs.struct), lv, stt)
    if st != 0:
        raise HmgpuError(st, "hmgpu_pack_levels")
    for k in range(3):
        out.struct.ctu_level_start[k] = out.starts[k].ctypes.data
    return out


class Context:
    """One hmgpu context = one GPU + one HIP stream + a pool of device pictures (HM: one TDecTop)."""

    def __init__(self, seq, device=0):
        self.seq = seq
        self.device = device
        self._h = C.c_void_p()
        st = lib().hmgpu_create(C.byref(seq), device, C.byref(self._h))
        if st != 0:
            raise HmgpuError(st, "hmgpu_create")
        self.num_ctus = lib().hmgpu_num_ctus(C.byref(seq))
        self.parts = lib().hmgpu_parts_per_ctu(C.byref(seq))

    def _chk(self, st, what):
        if st != 0:
            raise HmgpuError(st, what, lib().hmgpu_last_device_error(self._h))

    def close(self):
        if self._h:
            lib().hmgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def chroma_scale(self):
        """(log2 SubWidthC, log2 SubHeightC) of the context's chroma planes; monochrome keeps 4:2:0-shaped dummies"""
        f = self.seq.chroma_format
        return (0 if f == 3 else 1, 1 if f in (0, 1) else 0)

    def sync(self):
        self._chk(lib().hmgpu_sync(self._h), "hmgpu_sync")

    def debug_stall_intra(self, pic, ctu):
        """test hook: the intra wavefront leaves CTU `ctu` of the picture out (-1: off)"""
        self._chk(lib().hmgpu_debug_stall_intra(self._h, pic, ctu), "hmgpu_debug_stall_intra")

    # ---- pictures
    def acquire(self):
        h = C.c_int32(-1)
        self._chk(lib().hmgpu_picture_acquire(self._h, C.byref(h)), "hmgpu_picture_acquire")
        return h.value

    def release(self, pic):
        self._chk(lib().hmgpu_picture_release(self._h, pic), "hmgpu_picture_release")

    def upload(self, pic, planes):
        planes = [np.ascontiguousarray(p, dtype=np.int16) for p in planes]
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        strides = (C.c_int32 * 3)(*[p.shape[1] for p in planes])
        self._chk(lib().hmgpu_picture_upload(self._h, pic, ptrs, strides), "hmgpu_picture_upload")

    def download(self, pic):
        w, h = self.seq.width, self.seq.height
        sx, sy = self.chroma_scale
        planes = [np.zeros((h, w), dtype=np.int16), np.zeros((h >> sy, w >> sx), dtype=np.int16), np.zeros((h >> sy, w >> sx), dtype=np.int16)]
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        strides = (C.c_int32 * 3)(*[p.shape[1] for p in planes])
        self._chk(lib().hmgpu_picture_download(self._h, pic, ptrs, strides), "hmgpu_picture_download")
        return planes

    def set_streams(self, n):
        """lanes of replay(): 1 = serial kernels, 2 = two half-batches on two streams"""
        self._chk(lib().hmgpu_set_streams(self._h, n), "hmgpu_set_streams")

    def download_packed(self, pic, bytes_per_sample, crop=(0, 0, 0, 0)):
        """the picture as 8- or 16-bit planes cropped by (left, right, top, bottom) luma samples"""
        l, r, t, b = crop
        W, H = self.seq.width - l - r, self.seq.height - t - b
        dt = np.uint8 if bytes_per_sample == 1 else np.uint16
        sx, sy = self.chroma_scale
        planes = [np.zeros((H, W), dtype=dt), np.zeros((H >> sy, W >> sx), dtype=dt), np.zeros((H >> sy, W >> sx), dtype=dt)]
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        strides = (C.c_int32 * 3)(*[p.strides[0] for p in planes])
        self._chk(lib().hmgpu_picture_download_packed(self._h, pic, ptrs, strides, bytes_per_sample, l, r, t, b), "hmgpu_picture_download_packed")
        return planes

    def picture_hash(self, pic, method):
        """method 1 = MD5, 2 = CRC, 3 = checksum (decoded-picture-hash SEI); returns the bytes of Y, Cb, Cr concatenated"""
        dig = (C.c_uint8 * 48)()
        n = C.c_int32()
        self._chk(lib().hmgpu_picture_hash(self._h, pic, method, dig, C.byref(n)), "hmgpu_picture_hash")
        return np.array([dig[16 * k + i] for k in range(3) for i in range(n.value)], dtype=np.uint8)

    # ---- frame-parallel exchange (hmgpu.h: hmgpu_picture_device_region)
    def device_region(self, pic, receive=False):
        """(device address, bytes) of the contiguous plane region of `pic`: the finished picture, or where to receive one"""
        base, nbytes = C.c_void_p(), C.c_int64()
        self._chk(lib().hmgpu_picture_device_region(self._h, pic, 1 if receive else 0, C.byref(base), C.byref(nbytes)),
                  "hmgpu_picture_device_region")
        return int(base.value), int(nbytes.value)

    def commit_received(self, pic):
        self._chk(lib().hmgpu_picture_commit_received(self._h, pic), "hmgpu_picture_commit_received")

    def transfer_to(self, pic, other, other_pic):
        """the finished picture `pic` into picture `other_pic` of another context (same process; another GPU or this one)"""
        self._chk(lib().hmgpu_picture_transfer(self._h, pic, other._h, other_pic), "hmgpu_picture_transfer")

    @property
    def transfer_bytes(self):
        return int(lib().hmgpu_transfer_bytes(self._h))

    def stream_handle(self):
        return int(lib().hmgpu_stream(self._h) or 0)

    # ---- the two calls
    def decompress_slice(self, pic, slice_idx, slice_params, meta, coeffs, first_ctu=0, num_ctus=None):
        n = self.num_ctus - first_ctu if num_ctus is None else num_ctus
        st = lib().hmgpu_decompress_slice(self._h, pic, slice_idx, C.byref(slice_params), C.byref(meta.struct),
                                          C.byref(coeffs.struct), first_ctu, n)
        self._chk(st, "hmgpu_decompress_slice")

    def filter_picture(self, pic, pic_params, sao_raw=None, stages=7):
        arr = abi.sao_array_from_raw(sao_raw) if sao_raw is not None else None
        st = lib().hmgpu_filter_picture_stages(self._h, pic, C.byref(pic_params), arr, stages)
        self._chk(st, "hmgpu_filter_picture_stages")

    # ---- the same for several independent pictures per call
    @staticmethod
    def picture_jobs(jobs):
        """jobs: [(pic, [slice_params, ...], meta, coeffs), ...] with meta / coeffs anything that has .struct (MetaHolder, CoeffHolder)
        or a StagingHolder passed for both -> the C array (reusable; keeps what it points to alive)"""
        arr = (abi.PictureJob * len(jobs))()
        arr._keep = [jobs]
        for i, (pic, slices, meta, coeffs) in enumerate(jobs):
            sl = (C.POINTER(abi.SliceParams) * len(slices))(*[C.pointer(s) for s in slices])
            arr._keep.append(sl)
            arr[i].pic, arr[i].num_slices, arr[i].slices = pic, len(slices), sl
            arr[i].meta = C.pointer(meta.struct)
            arr[i].coeffs = C.pointer(coeffs.coeffs if isinstance(coeffs, abi.StagingHolder) else coeffs.struct)
        return arr

    @staticmethod
    def filter_jobs(jobs):
        """jobs: [(pic, pic_params, sao_array or None), ...]; sao_array = abi.sao_array_from_raw(...)"""
        arr = (abi.FilterJob * len(jobs))()
        arr._keep = [jobs]
        for i, (pic, pp, sao) in enumerate(jobs):
            arr[i].pic, arr[i].pp = pic, C.pointer(pp)
            arr[i].sao = C.cast(sao, C.c_void_p) if sao is not None else None
        return arr

    def decompress_pictures(self, jobs):
        arr = jobs if isinstance(jobs, C.Array) else self.picture_jobs(jobs)
        self._chk(lib().hmgpu_decompress_pictures(self._h, len(arr), arr), "hmgpu_decompress_pictures")

    def filter_pictures(self, jobs):
        arr = jobs if isinstance(jobs, C.Array) else self.filter_jobs(jobs)
        self._chk(lib().hmgpu_filter_pictures(self._h, len(arr), arr), "hmgpu_filter_pictures")

    def staging_alloc(self):
        h = C.c_void_p()
        m, co = abi.CtuMeta(), abi.Coeffs()
        self._chk(lib().hmgpu_staging_alloc(self._h, C.byref(h), C.byref(m), C.byref(co)), "hmgpu_staging_alloc")
        return abi.StagingHolder(h, m, co, self.num_ctus, abi.parts_per_ctu(self.seq), 1 << self.seq.log2_ctu_size)

    def pack_levels(self, meta, coeffs):
        return pack_levels(self.seq, meta, coeffs)

    def staging_free(self, st):
        lib().hmgpu_staging_free(self._h, st.handle)

    def replay(self, pics, stages, iters):
        pics = list(pics) if isinstance(pics, (list, tuple)) else [pics]
        arr = (C.c_int32 * len(pics))(*pics)
        self._chk(lib().hmgpu_replay_batch(self._h, arr, len(pics), stages, iters), "hmgpu_replay_batch")

    def set_profiling(self, on):
        self._chk(lib().hmgpu_set_profiling(self._h, int(on)), "hmgpu_set_profiling")

    def stats(self, reset=False):
        s = abi.Stats()
        self._chk(lib().hmgpu_get_stats(self._h, C.byref(s), int(reset)), "hmgpu_get_stats")
        out = {"intra_partitions": int(s.intra_partitions), "inter_partitions": int(s.inter_partitions), "kernels": {}}
        for k in range(abi.NUM_KERNELS):
            out["kernels"][lib().hmgpu_kernel_name(k).decode()] = (float(s.kernel_ms[k]), int(s.kernel_launches[k]))
        return out

    # ---- finer seams
    def inverse_transform_batch(self, levels, log2_size, bit_depth, qp_per, qp_rem, flags):
        levels = np.ascontiguousarray(levels, dtype=np.int16)
        n = levels.shape[0]
        per = np.ascontiguousarray(qp_per, dtype=np.int8)
        rem = np.ascontiguousarray(qp_rem, dtype=np.int8)
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        out = np.zeros_like(levels)
        st = lib().hmgpu_inverse_transform_batch(self._h, log2_size, bit_depth, n, _p(levels), _p(per), _p(rem), _p(fl), _p(out))
        self._chk(st, "hmgpu_inverse_transform_batch")
        return out

    def mc_batch(self, is_chroma, bit_depth, plane, blocks, bi):
        """blocks: int32 [n, 6] = x, y, w, h, mvx, mvy.  returns list of (h, w) arrays"""
        plane = np.ascontiguousarray(plane, dtype=np.int16)
        blocks = np.ascontiguousarray(blocks, dtype=np.int32)
        total = int((blocks[:, 2] * blocks[:, 3]).sum())
        dst = np.zeros(total, dtype=np.int16)
        st = lib().hmgpu_mc_batch(self._h, int(is_chroma), bit_depth, _p(plane), plane.shape[1], plane.shape[1], plane.shape[0],
                                  blocks.shape[0], _p(blocks), int(bi), _p(dst))
        self._chk(st, "hmgpu_mc_batch")
        out, pos = [], 0
        for b in blocks:
            w, h = int(b[2]), int(b[3])
            out.append(dst[pos:pos + w * h].reshape(h, w))
            pos += w * h
        return out
