#!/usr/bin/env python3
"""Bitstreams for the decoder benchmark (bench.py --workload decode): synthetic clips encoded by HM's own encoder (oracle/_ref), with
the decoded-picture-hash SEI (MD5) in the stream so that every decoded picture verifies itself -- no reference planes are stored.
Run in the BUILD container (needs /root/reference for the encoder build); takes several minutes per stream.
usage: python oracle/make_bench_stream.py [name ...]"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import hmref, make_golden      # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
STREAMS = {
    # name: (cfg, w, h, frames, bit depth, qp, extra)
    "bench_ldp_main10_3840x2160": ("encoder_lowdelay_P_main10.cfg", 3840, 2160, 5, 10, 32, []),
    # the same clip with wavefront parallel processing (one sub-stream per CTB row, entry points in the slice header): what x265-style
    # 4K streams look like, and what lets the parser work on several rows of one picture at once
    "bench_ldp_wpp_main10_3840x2160": ("encoder_lowdelay_P_main10.cfg", 3840, 2160, 5, 10, 32, ["--WaveFrontSynchro=1"]),
    "bench_ra_main10_1920x1080": ("encoder_randomaccess_main10.cfg", 1920, 1080, 9, 10, 32, ["--IntraPeriod=8"]),
    # a longer low-delay clip: one I picture and 16 small P pictures (the synthetic scene is nearly static)
    "bench_ldp_main10_1920x1080_17": ("encoder_lowdelay_P_main10.cfg", 1920, 1080, 17, 10, 32, []),
}


def main(names):
    for name in names or STREAMS:
        cfg, w, h, frames, bd, qp, extra = STREAMS[name]
        with tempfile.TemporaryDirectory() as tmp:
            yuv = os.path.join(tmp, "c.yuv")
            make_golden.write_yuv(yuv, make_golden.synth_clip(w, h, frames, bd, seed=2160 + frames, novel=True), bd)
            bs = os.path.join(OUT, name + ".bin")
            cmd = [hmref.ENCODER_PATH, "-c", os.path.join(make_golden.HM_CFG, cfg), "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "30",
                   "-f", str(frames), "--InputBitDepth=%d" % bd, "--InternalBitDepth=%d" % bd, "-q", str(qp), "-b", bs,
                   "-o", os.path.join(tmp, "rec.yuv"), "--SEIDecodedPictureHash=1", "--SearchRange=16", "--ECU=1", "--CFM=1", "--ESD=1"] + extra
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, timeout=3000)
            print("wrote %s (%d bytes)" % (bs, os.path.getsize(bs)))


if __name__ == "__main__":
    main(sys.argv[1:])
