#!/usr/bin/env python3
"""Manufacture the golden fixtures under tests/golden/ from the REAL HM 16.0 reference.

TEST INFRASTRUCTURE ONLY.  Runs in the builder container (needs /root/reference + `make -C oracle ref`).
The reference has no test vectors of its own, so every fixture is produced here by
running HM itself (oracle/_ref/TAppEncoder to make small bitstreams from seeded synthetic clips,
oracle/_ref/libhmref.so = HM's TDecTop + TLibCommon to decode them and to answer kernel-level KATs).
Fixtures are DATA only (inputs + HM's outputs); no reference source text is stored.

  python oracle/make_golden.py streams     # P1/P2: picture-level dumps of small real streams
  python oracle/make_golden.py kats        # K1/K3/K4: kernel-level known-answer tests
  python oracle/make_golden.py all
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import hmref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
HM_CFG = "/root/reference/cfg"


# ------------------------------------------------------------------------------------------ synthetic clips
def synth_clip(w, h, frames, bit_depth, seed, novel=False, fade=False, noisy=False, csx=1, csy=1):
    """gradient + checker + moving textured blobs + seeded noise, planar with chroma subsampled by (csx, csy), returns list of (Y,U,V) uint16"""
    rng = np.random.RandomState(seed)
    maxv = (1 << bit_depth) - 1
    big = 2 * max(w, h) + 128
    yy, xx = np.mgrid[0:big, 0:big].astype(np.float64)
    tex = (0.45 + 0.25 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 0.15 * (((xx // 16) + (yy // 16)) % 2)
           + 0.1 * np.sin((xx + 2 * yy) / 31.0))
    tex += 0.06 * rng.randn(big, big)
    tex2 = 0.5 + 0.3 * np.sin(xx / 5.0 + yy / 7.0) + 0.08 * rng.randn(big, big)
    out = []
    for f in range(frames):
        # global pan with sub-pel motion (bilinear resample) + one object moving the other way
        dx, dy = 1.75 * f, 0.5 * f
        x0, y0 = 40 + dx, 40 + dy
        xi, yi = int(np.floor(x0)), int(np.floor(y0))
        fx, fy = x0 - xi, y0 - yi
        a = tex[yi:yi + h + 1, xi:xi + w + 1]
        img = ((1 - fx) * (1 - fy) * a[:h, :w] + fx * (1 - fy) * a[:h, 1:w + 1]
               + (1 - fx) * fy * a[1:h + 1, :w] + fx * fy * a[1:h + 1, 1:w + 1])
        ox, oy = int(w * 0.3 - 3 * f), int(h * 0.35 + 2 * f)
        bw, bh = w // 4, h // 3
        x1, y1 = max(ox, 0), max(oy, 0)
        x2, y2 = min(ox + bw, w), min(oy + bh, h)
        if x2 > x1 and y2 > y1:
            img[y1:y2, x1:x2] = tex2[100 + y1 - oy:100 + y2 - oy, 100 + x1 - ox:100 + x2 - ox]
        if novel and f > 0:
            # content with no counterpart in any earlier frame (a new smooth ramp + stripes patch per frame): intra CUs in P pictures
            pw_, ph_ = 56, 48
            px_, py_ = int(w * 0.5) + 9 * f, int(h * 0.08) + 5 * f
            gy, gx = np.mgrid[0:ph_, 0:pw_].astype(np.float64)
            img[py_:py_ + ph_, px_:px_ + pw_] = 0.5 + 0.4 * np.sin(gx / (2.0 + f) + f) * np.cos(gy / (3.0 + 0.5 * f)) * (1 if f % 2 else -1)
        img = img + 0.012 * rng.randn(h, w)
        if noisy:                           # a patch of full-range white noise (new every frame): cheaper to send raw than to predict
            img[h // 4:h // 4 + 64, w // 3:w // 3 + 96] = rng.rand(64, 96)
        if fade:                            # brightness ramps from frame to frame: what explicit weighted prediction is for
            img = img * (1.0 - 0.09 * f) + 0.02 * f
        Y = np.clip(np.round(img * maxv), 0, maxv).astype(np.uint16)
        ch, cw = h >> csy, w >> csx
        # (chroma follows the luma texture in part: something for cross-component prediction to find in 4:4:4)
        ysub = Y[::1 << csy, ::1 << csx].astype(np.float64) / maxv - 0.5
        U = np.clip(np.round((0.5 + 0.2 * np.sin((xx[:ch, :cw] * (2 >> (1 - csx)) / 2.0 + 3 * f) / 11.0) + (0.0 if csx else 0.15 * ysub)) * maxv), 0, maxv).astype(np.uint16)
        V = np.clip(np.round((0.5 + 0.2 * np.cos((yy[:ch, :cw] * (2 >> (1 - csy)) / 2.0 - 2 * f) / 17.0)
                              + 0.1 * ysub) * maxv), 0, maxv).astype(np.uint16)
        out.append((Y, U, V))
    return out


def write_yuv(path, clip, bit_depth, mono=False):
    with open(path, "wb") as f:
        for planes in clip:
            for p in (planes[:1] if mono else planes):
                f.write(p.astype(np.uint8 if bit_depth == 8 else "<u2").tobytes())


# ------------------------------------------------------------------------------------------ streams
STREAMS = {
    # name: (cfg, w, h, frames, input bit depth, internal bit depth, qp, extra encoder args)
    "ldp_main8_416x240": ("encoder_lowdelay_P_main.cfg", 416, 240, 3, 8, 8, 32, []),
    "ra_main10_208x120": ("encoder_randomaccess_main10.cfg", 208, 120, 9, 8, 10, 30, ["--IntraPeriod=8"]),
    "ldp_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 4, 10, 10, 26, []),
    # NOTE: multi-slice inter streams (--SliceMode=1) are not used: HM 16.0's own TAppDecoder asserts
    # (TComBitStream.h:191) on the streams its encoder writes for them, at every optimisation level.
    "intra_main10_208x120": ("encoder_intra_main10.cfg", 208, 120, 1, 10, 10, 30, []),
    # 3 x 2 tiles (HEVC wants tiles >= 256 x 64), no loop filtering across tile borders: tile-bounded intra references, deblocking and SAO neighbourhoods
    "ldp_tiles_main10_832x128": ("encoder_lowdelay_P_main10.cfg", 832, 128, 3, 10, 10, 30,
                                 ["--TileUniformSpacing=1", "--NumTileColumnsMinus1=2", "--NumTileRowsMinus1=1", "--LFCrossTileBoundaryFlag=0"]),
    # lossless CUs (cu_transquant_bypass_flag forced: HM's RD never picks it on its own here); PCM CUs on a noisy patch, chosen by
    # RD in I and P pictures, with PCM samples exempt from the loop filters
    "ldp_lossless_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 30, ["--TransquantBypassEnableFlag=1", "--CUTransquantBypassFlagForce=1"]),
    "ldp_pcm_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 6,
                              ["--PCMEnabledFlag=1", "--PCMLog2MaxSize=5", "--PCMLog2MinSize=3", "--PCMFilterDisableFlag=1",
                               "--DeblockingFilterControlPresent=1", "--LoopFilterOffsetInPPS=1", "--LoopFilterBetaOffset_div2=6", "--LoopFilterTcOffset_div2=6"]),
    # scaling lists from a file: every matrix different, DC values for 16x16 / 32x32 (I picture: intra lists, P pictures: inter lists)
    "ldp_sl_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 28, ["--ScalingList=2", "--ScalingListFile=@SLFILE@"]),
    "ldp_sldef_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 2, 8, 8, 34, ["--ScalingList=1"]),
    # explicit weighted prediction on a fading clip: P slices (weighted_pred_flag) and B slices (weighted_bipred_flag)
    "ldp_wp_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 4, 10, 10, 30, ["--WeightedPredP=1"]),
    "ra_wp_main8_208x120": ("encoder_randomaccess_main.cfg", 208, 120, 5, 8, 8, 32, ["--IntraPeriod=8", "--WeightedPredP=1", "--WeightedPredB=1"]),
    # constrained intra prediction (inter neighbours are not intra references) and no strong smoothing, P pictures with intra CUs
    "ldp_cip_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 30, ["--ConstrainedIntraPred=1", "--StrongIntraSmoothing=0"]),
}


# 4:2:2 and 4:4:4: the range-extension configurations with the chroma format switched; 4:4:4 with and without
# cross-component prediction, inter and intra, a bit depth of 10 in one of each
CF444 = ["--InputChromaFormat=444", "--ChromaFormatIDC=444"]
CF422 = ["--InputChromaFormat=422", "--ChromaFormatIDC=422", "--CrossComponentPrediction=0"]
STREAMS.update({
    "ldb_444_ccp_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 8, 27, CF444),
    "intra_444_ccp_main10_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 10, 10, 24, CF444),
    "ldb_422_main10_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 10, 10, 28, CF422),
    "intra_422_main8_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 8, 8, 24, CF422),
})


def write_scaling_list_file(path):
    """a scaling list file in HM's text format (TComScalingList::xParseScalingList, TComSlice.cpp:2052-2138; names TComRom.cpp:580-638)
    with a different seeded matrix for every size and list"""
    rng = np.random.RandomState(0x5CA1E)
    names = {0: "4X4", 1: "8X8", 2: "16X16", 3: "32X32"}
    lists = ["INTRA%s_LUMA", "INTRA%s_CHROMAU", "INTRA%s_CHROMAV", "INTER%s_LUMA", "INTER%s_CHROMAU", "INTER%s_CHROMAV"]
    with open(path, "w") as f:
        for sz in range(4):
            n = 4 if sz == 0 else 8
            for l, pat in enumerate(lists):
                if sz == 3 and l % 3:
                    continue                                       # 32x32 chroma is derived from 16x16 chroma
                base = pat % names[sz]
                yy, xx = np.mgrid[0:n, 0:n]
                m = np.clip(10 + 3 * (xx + yy) + rng.randint(-3, 4, size=(n, n)) + 2 * l, 4, 200)
                f.write(base + "\n")
                for r in range(n):
                    f.write(",".join(str(int(v)) for v in m[r]) + ",\n")
                if sz >= 2:
                    f.write(base + "_DC\n%d\n" % int(rng.randint(8, 40)))
                f.write("\n")


def chroma_scale_of(name):
    """(csx, csy) of a stream by its name: _444_ / _422_, else 4:2:0"""
    return (0, 0) if "_444" in name else (1, 0) if "_422" in name else (1, 1)


def split_planes(fr, w, h, csx, csy):
    cw, ch = w >> csx, h >> csy
    return fr[:w * h].reshape(h, w), fr[w * h:w * h + cw * ch].reshape(ch, cw), fr[w * h + cw * ch:w * h + 2 * cw * ch].reshape(ch, cw)


# bit depth 12 in 4:2:0: interpolation head room 2 (TComInterpolationFilter.cpp:195-212), transformShift = 15 - 12 - log2(N)
# negative for 32x32 blocks (TComTrQuant.cpp:1203-1313), QP range up to 51 + 24
CF420 = ["--InputChromaFormat=420", "--ChromaFormatIDC=420", "--CrossComponentPrediction=0", "--ExtendedPrecision=0"]
STREAMS.update({
    "ldb_main12_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 26, CF420),
    "intra_main12_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 12, 12, 22, CF420),
})


def encode(name, tmp):
    cfg, w, h, frames, ibd, bd, qp, extra = STREAMS[name]
    yuv = os.path.join(tmp, name + ".yuv")
    csx, csy = chroma_scale_of(name)
    clip = synth_clip(w, h, frames, ibd, seed=0x484D + sum(map(ord, name)), novel="cip" in name, fade="wp" in name, noisy="pcm" in name or "lossless" in name,
                      csx=csx, csy=csy)
    mono = "mono" in name                                               # 4:0:0: luma only in, luma only out
    write_yuv(yuv, clip, ibd, mono)
    bs = os.path.join(tmp, name + ".bin")
    rec = os.path.join(tmp, name + "_rec.yuv")
    if any("@SLFILE@" in e for e in extra):
        slf = os.path.join(tmp, name + "_sl.txt")
        write_scaling_list_file(slf)
        extra = [e.replace("@SLFILE@", slf) for e in extra]
    cfg_lines = [e[4:] for e in extra if e.startswith("cfg:")]          # options that only parse from a configuration file (arrays)
    extra = [e for e in extra if not e.startswith("cfg:")]
    if cfg_lines:
        xcfg = os.path.join(tmp, name + "_extra.cfg")
        with open(xcfg, "w") as f:
            f.write("\n".join(cfg_lines) + "\n")
        extra = ["-c", xcfg] + extra
    cmd = [hmref.ENCODER_PATH, "-c", os.path.join(HM_CFG, cfg), "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "30",
           "-f", str(frames), "--InputBitDepth=%d" % ibd, "--InternalBitDepth=%d" % bd, "--OutputBitDepth=%d" % bd,
           "-q", str(qp), "-b", bs, "-o", rec, "--SEIDecodedPictureHash=1", "--Level=3.1"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)   # (some slice/tile option mixes hang HM's encoder)
    if r.returncode != 0:
        print(r.stdout[-3000:])
        raise RuntimeError("encoder failed for " + name)
    with open(bs, "rb") as f:
        data = f.read()
    # (the reconstruction file has 16-bit samples as soon as either component has more than 8 bits)
    per8 = w * h if mono else w * h + 2 * (w >> csx) * (h >> csy)
    recdata = np.fromfile(rec, dtype="<u2" if os.path.getsize(rec) == 2 * per8 * frames else np.uint8)
    return data, recdata, (w, h, frames, bd)


def dump_stream(name, bitstream, enc_rec, geom):
    w, h, frames, bd = geom
    dec = hmref.RefDecoder(bitstream, check_hash=True)
    out = {"bitstream": np.frombuffer(bitstream, dtype=np.uint8)}
    pics = []
    final_by_poc = {}
    while dec.next():
        info = dec.info()
        k = "pic%02d_" % len(pics)
        ns = info["num_slices"]
        out[k + "info"] = np.array([info[x] for x in ["width", "height", "bd_y", "bd_c", "poc", "slice_type", "num_ctus",
                                                      "ctus_w", "parts", "ctu_size", "num_slices", "use_sao",
                                                      "lf_across_tiles", "chroma_format", "tid", "max_depth"]], dtype=np.int32)
        out[k + "slices"] = dec.slices(ns)
        out[k + "wp"] = dec.wp(ns)
        out[k + "scaling_lists"] = dec.scaling_lists()
        out[k + "tile_idx"] = dec.tile_idx(info["num_ctus"])
        meta = dec.meta(info)
        for n2, a in meta.items():
            out[k + "meta_" + n2] = a
        pcm_info, pcm = dec.pcm(info)
        out[k + "pcm_info"] = pcm_info
        if pcm_info[2] and np.any(meta["ipcm"]):                  # PCM samples only where PCM CUs exist (they are large)
            for c in range(3):
                out[k + "pcm%d" % c] = pcm[c]
        if info["chroma_format"] == 3:
            for c, a in enumerate(dec.ccp_alpha(info)):
                out[k + "meta_ccp_" + "uv"[c]] = a
        co = dec.coeffs(info)
        for c in range(3):
            assert co[c].min() >= -32768 and co[c].max() <= 32767
            out[k + "coeff%d" % c] = co[c].astype(np.int16)
        out[k + "sao_raw"] = dec.sao_params(info)
        for c, p in enumerate(dec.planes(info)):
            out[k + "pre%d" % c] = p
        dec.filter_step()
        for c, p in enumerate(dec.planes(info)):
            out[k + "dbk%d" % c] = p
        dec.filter_step()
        out[k + "sao_rec"] = dec.sao_params(info)
        fin = dec.planes(info)
        for c, p in enumerate(fin):
            out[k + "fin%d" % c] = p
        out[k + "crc"], out[k + "checksum"] = dec.hashes()
        ok, md5 = dec.finish()
        assert ok, "HM hash mismatch in %s pic %d" % (name, len(pics))
        out[k + "md5"] = md5
        final_by_poc[info["poc"]] = fin
        pics.append(info)
        print("  %s pic %d: POC %d type %d slices %d" % (name, len(pics) - 1, info["poc"], info["slice_type"], ns))
    dec.close()
    assert len(pics) == frames
    # encoder reconstruction == decoder output (HM practice): check in POC order
    csx, csy = chroma_scale_of(name)
    per = w * h + 2 * (w >> csx) * (h >> csy)
    for poc in sorted(final_by_poc):
        fr = enc_rec[poc * per:(poc + 1) * per].astype(np.int16)
        for a, b in zip(split_planes(fr, w, h, csx, csy), final_by_poc[poc]):
            assert np.array_equal(a, b)
    out["num_pics"] = np.array([len(pics)], dtype=np.int32)
    path = os.path.join(GOLD, "stream_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


# ------------------------------------------------------------------------------------------ "lite" streams
# Streams whose syntax HM 16.0's own decoder does not get through (it asserts in TComBitStream.h:191 on the multi-slice / WPP
# streams its encoder writes) or that add nothing to the metadata fixtures: the bitstream plus the ENCODER's reconstruction, which
# is by HM practice what a conforming decoder must output, and whose MD5 the encoder put into the stream as SEI.
REXT420 = ["--InputChromaFormat=420", "--ChromaFormatIDC=420", "--CrossComponentPrediction=0", "--HighPrecisionPredictionWeighting=0",
           "--TransformSkipLog2MaxSize=2"]
MONO = ["--InputChromaFormat=400", "--ChromaFormatIDC=400", "--CrossComponentPrediction=0", "--HighPrecisionPredictionWeighting=0", "--TransformSkipLog2MaxSize=2"]
LITE = {
    # name: (cfg, w, h, frames, input bit depth, internal bit depth, qp, extra encoder args)
    # (HM 16.0's encoder writes every slice NAL with the data of all following CTUs of the picture appended unless slice SEGMENTS
    # are configured as well -- the reason its own decoder trips over them; with SliceSegmentArgument == SliceArgument every slice
    # is one independent segment and the stream is well formed)
    "ldp_slices_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 32,
                                 ["--SliceMode=1", "--SliceArgument=3", "--SliceSegmentMode=1", "--SliceSegmentArgument=3", "--LFCrossSliceBoundaryFlag=0"]),
    "ldp_depslices_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 30,
                                     ["--SliceMode=1", "--SliceArgument=4", "--SliceSegmentMode=1", "--SliceSegmentArgument=2", "--LFCrossSliceBoundaryFlag=1"]),
    "ldp_wpp_main10_416x240": ("encoder_lowdelay_P_main10.cfg", 416, 240, 3, 10, 10, 32, ["--WaveFrontSynchro=1"]),
    "ldp_wpp_depslices_main8_416x240": ("encoder_lowdelay_P_main.cfg", 416, 240, 3, 8, 8, 34,
                                        ["--WaveFrontSynchro=1", "--SliceMode=1", "--SliceArgument=10", "--SliceSegmentMode=1", "--SliceSegmentArgument=3"]),
    "ldp_dqp_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 30, ["--MaxDeltaQP=3", "--MaxCuDQPDepth=2"]),
    "ra_cra_main8_208x120": ("encoder_randomaccess_main.cfg", 208, 120, 18, 8, 8, 34, ["--IntraPeriod=8", "--DecodingRefreshType=1"]),
    "ldp_ctu32_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 30, ["--MaxCUWidth=32", "--MaxCUHeight=32", "--MaxPartitionDepth=3"]),
    "ldp_ctu16_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 30, ["--MaxCUWidth=16", "--MaxCUHeight=16", "--MaxPartitionDepth=2", "--QuadtreeTULog2MaxSize=4"]),
    "ldp_crop_main8_204x116": ("encoder_lowdelay_P_main.cfg", 204, 116, 3, 8, 8, 32, ["--ConformanceWindowMode=1"]),
    "ldp_tileslices_main10_832x128": ("encoder_lowdelay_P_main10.cfg", 832, 128, 3, 10, 10, 32,
                                      ["--TileUniformSpacing=1", "--NumTileColumnsMinus1=2", "--NumTileRowsMinus1=1", "--SliceMode=3", "--SliceArgument=2",
                                       "--SliceSegmentMode=1", "--SliceSegmentArgument=64", "--LFCrossSliceBoundaryFlag=0"]),
    "ldb_main8_208x120": ("encoder_lowdelay_main.cfg", 208, 120, 4, 8, 8, 32, []),
    # non-zero PPS chroma QP offsets (chroma dequantisation and chroma deblocking QP), VUI in the SPS, access unit delimiters,
    # cabac_init_flag in use, three merge candidates
    "ldp_cqp_vui_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 4, 10, 10, 30,
                                   ["--CbQpOffset=4", "--CrQpOffset=-3", "--VuiParametersPresent=1", "--AspectRatioInfoPresent=1", "--AspectRatioIdc=255",
                                    "--SarWidth=4", "--SarHeight=3", "--AccessUnitDelimiter=1", "--CabacInitPresent=1", "--MaxNumMergeCand=3"]),
    # the loop-filter combinations the fused kernel does not serve: no SAO (deblocking only), no deblocking (SAO only), neither
    "ldp_nosao_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 32, ["--SAO=0"]),
    "ldp_nodbk_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 32, ["--LoopFilterDisable=1"]),
    "ldp_nofilters_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 32, ["--SAO=0", "--LoopFilterDisable=1"]),
    # no temporal motion vector prediction; 16x16 minimum CUs with 8x8 minimum transform blocks and a deeper transform tree
    "ra_notmvp_main8_208x120": ("encoder_randomaccess_main.cfg", 208, 120, 9, 8, 8, 32, ["--IntraPeriod=8", "--TMVPMode=0"]),
    "ldp_mincu16_main10_208x112": ("encoder_lowdelay_P_main10.cfg", 208, 112, 3, 10, 10, 28,
                                   ["--MaxPartitionDepth=3", "--QuadtreeTULog2MinSize=3", "--QuadtreeTUMaxDepthInter=2", "--QuadtreeTUMaxDepthIntra=2"]),
    # fine quantisation, intra only: transform skip and sign data hiding at work on every 4x4 block
    "intra_qp12_main8_208x120": ("encoder_intra_main.cfg", 208, 120, 2, 8, 8, 12, []),
    # parsing / derivation rules that only certain parameter values reach
    "ra_parmrg4_main8_208x120": ("encoder_randomaccess_main.cfg", 208, 120, 5, 8, 8, 32, ["--IntraPeriod=8", "--Log2ParallelMergeLevel=4"]),
    "ldp_tudepth1_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 30, ["--QuadtreeTUMaxDepthInter=1", "--QuadtreeTUMaxDepthIntra=1"]),
    "ldp_maxtb16_noamp_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 30, ["--QuadtreeTULog2MaxSize=4", "--AMP=0"]),
    "ldp_nots_nosdh_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 26, ["--TransformSkip=0", "--SignHideFlag=0"]),
    "ldp_ctu32_mincu16_main8_224x128": ("encoder_lowdelay_P_main.cfg", 224, 128, 3, 8, 8, 30, ["--MaxCUWidth=32", "--MaxCUHeight=32", "--MaxPartitionDepth=2"]),
    "ldp_qpneg_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 2, 10, 10, -8, []),
    "ldp_qp48_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 4, 8, 8, 48, []),
    "ldp_slicedbk_main10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 10, 10, 34,
                                    ["--DeblockingFilterControlPresent=1", "--LoopFilterOffsetInPPS=0", "--LoopFilterBetaOffset_div2=-3", "--LoopFilterTcOffset_div2=4"]),
    "ldp_qgctu_main8_208x120": ("encoder_lowdelay_P_main.cfg", 208, 120, 3, 8, 8, 30, ["--MaxDeltaQP=2", "--MaxCuDQPDepth=0"]),
    # different bit depths for luma and chroma (10 / 8 and 8 / 10), extreme chroma QP offsets
    "ldp_bd10_8_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 8, 10, 30, ["--InternalBitDepthC=8", "--CbQpOffset=-12", "--CrQpOffset=12"]),
    "ldp_bd8_10_208x120": ("encoder_lowdelay_P_main10.cfg", 208, 120, 3, 8, 8, 30, ["--InternalBitDepthC=10", "--CbQpOffset=12", "--CrQpOffset=-12"]),
    "ldp_tilesexp_main10_832x192": ("encoder_lowdelay_P_main10.cfg", 832, 192, 2, 10, 10, 34,
                                    ["--TileUniformSpacing=0", "--NumTileColumnsMinus1=2", "--TileColumnWidthArray=4,5", "--NumTileRowsMinus1=1", "--TileRowHeightArray=1", "--LFCrossTileBoundaryFlag=1"]),
    # range-extension coding tools that exist in 4:2:0 (rotation, implicit + explicit RDPCM, the single significance context,
    # persistent Rice adaptation); cross-component prediction needs 4:4:4 and stays off
    "ldb_rext420_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 8, 27, REXT420),
    "ldb_rext420_lossless_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 2, 8, 8, 30,
                                           REXT420 + ["--TransquantBypassEnableFlag=1", "--CUTransquantBypassFlagForce=1"]),
    "intra_rext420_main8_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 2, 8, 8, 24, REXT420),
    "intra_rext420_lossless_main8_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 8, 8, 30,
                                             REXT420 + ["--TransquantBypassEnableFlag=1", "--CUTransquantBypassFlagForce=1"]),
    # transform skip up to 32x32 without intra reference smoothing; weighted prediction with offsets in units of the bit depth
    # (log2_sao_offset_scale can only be non-zero above 10 bits: outside the device's bit depths)
    "ldb_rext420_ts32_nosmooth_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 8, 27,
                                                REXT420[:-1] + ["--TransformSkipLog2MaxSize=5", "--IntraReferenceSmoothing=0"]),
    "ldb_rext420_wp_hp_main10_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 10, 10, 30,
                                                  REXT420[:-2] + ["--TransformSkipLog2MaxSize=2", "--WeightedPredP=1", "--WeightedPredB=1", "--HighPrecisionPredictionWeighting=1"]),
    # the parser state of the range extensions across the places contexts are stored and restored: wavefronts with dependent slice
    # segments (StatCoeff synchronised with the context variables), slices of tiles (reset per tile)
    "ldb_rext420_wpp_depslices_main8_416x240": ("encoder_lowdelay_main_rext.cfg", 416, 240, 3, 8, 8, 26,
                                                REXT420 + ["--WaveFrontSynchro=1", "--SliceMode=1", "--SliceArgument=10", "--SliceSegmentMode=1", "--SliceSegmentArgument=3"]),
    "ldb_rext420_tileslices_main10_832x128": ("encoder_lowdelay_main_rext.cfg", 832, 128, 3, 10, 10, 28,
                                              REXT420 + ["--TileUniformSpacing=1", "--NumTileColumnsMinus1=2", "--NumTileRowsMinus1=1", "--SliceMode=3", "--SliceArgument=2",
                                                         "--SliceSegmentMode=1", "--SliceSegmentArgument=64", "--LFCrossSliceBoundaryFlag=0"]),
    # monochrome (chroma_format_idc 0): no chroma syntax anywhere, one digest in the hash SEI; with the range-extension tools, with
    # weighted prediction, SAO and a conformance window in luma units, and all intra
    "ldb_mono_rext_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 8, 30, MONO),
    "ldb_mono_wp_crop_main10_204x116": ("encoder_lowdelay_main_rext.cfg", 204, 116, 3, 10, 10, 32,
                                        MONO + ["--WeightedPredP=1", "--WeightedPredB=1", "--ConformanceWindowMode=1"]),
    "intra_mono_main8_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 2, 8, 8, 26, MONO),
    "ldb_rext420_mixed_main10_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 10, 10, 32,
                                         REXT420 + ["--TransquantBypassEnableFlag=1", "--CostMode=mixed_lossless_lossy"]),
}


# 4:2:2 / 4:4:4 variants: bitstream + the encoder's reconstruction
LITE.update({
    "ldb_444_main10_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 10, 10, 30, CF444 + ["--CrossComponentPrediction=0"]),
    "ldb_444_lossless_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 2, 8, 8, 30, CF444 + ["--TransquantBypassEnableFlag=1", "--CUTransquantBypassFlagForce=1"]),
    "intra_444_ts32_nosmooth_main8_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 8, 8, 27, CF444 + ["--TransformSkipLog2MaxSize=5", "--IntraReferenceSmoothing=0"]),
    # cross-component prediction between a 10-bit luma and an 8-bit chroma residual (getDifferentialLumaChromaBitDepth)
    "ldb_444_ccp_bd10_8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 10, 28, CF444 + ["--InternalBitDepthC=8"]),
    "ldb_444_ctu16_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 8, 30, CF444 + ["--MaxCUWidth=16", "--MaxCUHeight=16", "--MaxPartitionDepth=2", "--QuadtreeTULog2MaxSize=4"]),
    "ldb_422_lossless_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 2, 8, 8, 30, CF422 + ["--TransquantBypassEnableFlag=1", "--CUTransquantBypassFlagForce=1"]),
    "ldb_422_wp_main10_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 10, 10, 30, CF422 + ["--WeightedPredP=1", "--WeightedPredB=1"]),
    "ldb_422_wpp_depslices_main8_416x240": ("encoder_lowdelay_main_rext.cfg", 416, 240, 3, 8, 8, 28,
                                            CF422 + ["--WaveFrontSynchro=1", "--SliceMode=1", "--SliceArgument=10", "--SliceSegmentMode=1", "--SliceSegmentArgument=3"]),
    "intra_422_qp12_main10_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 10, 10, 12, CF422),
    "ldb_422_ctu32_main8_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 8, 8, 30, CF422 + ["--MaxCUWidth=32", "--MaxCUHeight=32", "--MaxPartitionDepth=3"]),
})


LITE.update({
    # 12 bits with what moves with the bit depth: transform skip up to 32x32 (negative transform shift: a left shift, TComTrQuant.cpp:1920-1959),
    # explicit weighted prediction (shift 14 - 12), scaling lists, PCM at a smaller bit depth, a low QP (large levels) and a 10-bit chroma
    "ldb_ts32_main12_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 24, CF420 + ["--TransformSkipLog2MaxSize=5"]),
    "ldb_wp_main12_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 28, CF420 + ["--WeightedPredP=1", "--WeightedPredB=1"]),
    "ldb_sl_main12_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 27, CF420 + ["--ScalingList=2", "--ScalingListFile=@SLFILE@"]),
    "intra_qp4_main12_208x120": ("encoder_intra_main_rext.cfg", 208, 120, 1, 12, 12, 4, CF420),
    "ldb_bd12_10_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 28, CF420 + ["--InternalBitDepthC=10"]),
    # 12 bits at the other chroma formats (4:4:4 with cross-component prediction)
    "ldb_444_ccp_main12_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 27, CF444 + ["--ExtendedPrecision=0"]),
    "ldb_422_main12_208x120": ("encoder_lowdelay_main_rext.cfg", 208, 120, 3, 12, 12, 28, CF422 + ["--ExtendedPrecision=0"]),
})
LITE_BD12 = ["ldb_ts32_main12_208x120", "ldb_wp_main12_208x120", "ldb_sl_main12_208x120", "intra_qp4_main12_208x120", "ldb_bd12_10_208x120",
             "ldb_444_ccp_main12_208x120", "ldb_422_main12_208x120"]


def make_lite(names=None):
    os.makedirs(GOLD, exist_ok=True)
    STREAMS.update(LITE)
    with tempfile.TemporaryDirectory() as tmp:
        for name in (names or LITE):
            print("encoding", name)
            bs, rec, (w, h, frames, bd) = encode(name, tmp)
            out = {"bitstream": np.frombuffer(bs, dtype=np.uint8), "geom": np.array([w, h, frames, bd], dtype=np.int32)}
            mono = "mono" in name
            csx, csy = chroma_scale_of(name)
            per = w * h if mono else w * h + 2 * (w >> csx) * (h >> csy)
            assert rec.size == per * frames
            for poc in range(frames):                                      # the recon file is in output (POC) order, cropped
                fr = rec[poc * per:(poc + 1) * per].astype(np.int16)
                out["poc%02d_0" % poc] = fr[:w * h].reshape(h, w)
                if mono:
                    continue
                _, out["poc%02d_1" % poc], out["poc%02d_2" % poc] = split_planes(fr, w, h, csx, csy)
            path = os.path.join(GOLD, "lite_%s.npz" % name)
            np.savez_compressed(path, **out)
            print("wrote %s (%.1f KB, %d bytes of bitstream)" % (path, os.path.getsize(path) / 1024.0, len(bs)))


def make_streams(names=None):
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for name in (names or STREAMS):
            print("encoding", name)
            bs, rec, geom = encode(name, tmp)
            dump_stream(name, bs, rec, geom)


# ------------------------------------------------------------------------------------------ KATs
def xorshift_rng(seed):
    return np.random.RandomState(seed & 0x7FFFFFFF)


def make_kats():
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    # K1: inverse transforms through HM xITrMxN ------------------------------------------------
    for bd in (8, 10):
        hmref.kat_init(bd, bd)
        for n in (4, 8, 16, 32):
            rng = xorshift_rng(0x484D3136 + n * 16 + bd)
            blocks = []
            blocks.append(rng.randint(-32768, 32768, size=(6, n, n)))                       # full range stress
            imp = np.zeros((8, n, n), dtype=np.int64)                                       # single-basis impulses
            for i, (r, c, v) in enumerate([(0, 0, 32767), (0, 0, -32768), (n - 1, n - 1, 32767), (n - 1, 0, -32768),
                                           (0, n - 1, 1), (1, 2 % n, -1), (n // 2, n // 2, 32767), (3 % n, 1, 1000)]):
                imp[i, r, c] = v
            blocks.append(imp)
            lowf = np.zeros((6, n, n), dtype=np.int64)                                      # typical low-frequency
            k = min(n, 8)
            lowf[:, :k, :k] = np.round(rng.laplace(0, 120, size=(6, k, k))) * (rng.rand(6, k, k) < 0.35)
            blocks.append(np.clip(lowf, -32768, 32767))
            coeff = np.concatenate(blocks).astype(np.int32)
            out["itr_in_n%d_bd%d" % (n, bd)] = coeff.astype(np.int16)
            out["itr_dct_n%d_bd%d" % (n, bd)] = hmref.kat_itr(bd, coeff, 0).astype(np.int16)
            if n == 4:
                out["itr_dst_n4_bd%d" % bd] = hmref.kat_itr(bd, coeff, 1).astype(np.int16)
    # K3: interpolation through HM TComInterpolationFilter ---------------------------------------
    for bd in (8, 10):
        hmref.kat_init(bd, bd)
        rng = xorshift_rng(0x1234 + bd)
        plane = rng.randint(0, 1 << bd, size=(96, 96)).astype(np.int16)
        plane[40:56, 40:56] = (1 << bd) - 1      # saturated patch: exercises the final clip
        plane[20:30, 60:70] = 0
        out["interp_plane_bd%d" % bd] = plane
        cases = []
        res = []
        for comp, nfrac in ((0, 4), (1, 8)):
            sizes = [(8, 8), (16, 4), (4, 16), (32, 32), (12, 16)] if comp == 0 else [(4, 4), (8, 2), (2, 8), (16, 16), (6, 8)]
            for bi in (0, 1):
                for yf in range(nfrac):
                    for xf in range(nfrac):
                        w, h = sizes[(xf + yf * nfrac + bi) % len(sizes)]
                        x0 = 8 + int(rng.randint(0, 96 - 16 - w))
                        y0 = 8 + int(rng.randint(0, 96 - 16 - h))
                        d = hmref.kat_interp(comp, plane, x0, y0, w, h, xf, yf, bi)
                        cases.append([comp, bi, xf, yf, x0, y0, w, h])
                        res.append(d.ravel())
        out["interp_cases_bd%d" % bd] = np.array(cases, dtype=np.int32)
        out["interp_out_bd%d" % bd] = np.concatenate(res).astype(np.int16)
        # addAvg on 14-bit intermediates
        a = rng.randint(-8192, 8192 + (1 << 13), size=(16, 16)).astype(np.int16)
        b = rng.randint(-8192, 8192 + (1 << 13), size=(16, 16)).astype(np.int16)
        out["addavg_a_bd%d" % bd] = a
        out["addavg_b_bd%d" % bd] = b
        out["addavg_out_bd%d" % bd] = hmref.kat_addavg(a, b)
    # K4: SAO offsetBlock -------------------------------------------------------------------------
    for bd in (8, 10):
        rng = xorshift_rng(0x5A0 + bd)
        maxo = 7 if bd == 8 else 31
        # smooth-ish plane so that all five edge classes occur
        base = rng.randint(0, 1 << bd, size=(40, 48)).astype(np.int64)
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 0)) // 4
        base[:, :6] = (1 << bd) - 1 - (base[:, :6] % 3)       # near-max: clip at the top
        base[:5, :] = base[:5, :] % 3                          # near-zero: clip at the bottom
        plane = base.astype(np.int16)
        out["sao_plane_bd%d" % bd] = plane
        cases, res = [], []
        for comp in (0, 1):
            for typ in range(5):
                for av in range(0, 256, 5 if typ in (2, 3) else 37):
                    avail = [(av >> i) & 1 for i in range(8)]
                    off = np.zeros(32, dtype=np.int32)
                    if typ == 4:
                        start = int(rng.randint(0, 32))
                        for i in range(4):
                            off[(start + i) % 32] = int(rng.randint(-maxo, maxo + 1))
                    else:
                        off[0], off[1] = int(rng.randint(0, maxo + 1)), int(rng.randint(0, maxo + 1))
                        off[3], off[4] = -int(rng.randint(0, maxo + 1)), -int(rng.randint(0, maxo + 1))
                    w, h = (32, 24) if (av & 1) else (17, 9)
                    x0, y0 = 4, 4
                    r = hmref.kat_sao_block(comp, bd, bd, typ, off, plane, x0, y0, w, h, avail)
                    cases.append([comp, typ, av, x0, y0, w, h] + list(off))
                    res.append(r[y0:y0 + h, x0:x0 + w].ravel())
        out["sao_cases_bd%d" % bd] = np.array(cases, dtype=np.int32)
        out["sao_out_bd%d" % bd] = np.concatenate(res).astype(np.int16)
    path = os.path.join(GOLD, "kats.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("kats", "all"):
        make_kats()
    if what in ("streams", "all"):
        make_streams(sys.argv[2:] or None)
    if what in ("lite", "all"):
        make_lite(sys.argv[2:] or None)
