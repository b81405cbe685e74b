#!/usr/bin/env python3
"""CPU baseline calibration (CPU baseline plan 1 + 3), run in the BUILD container (needs /root/reference):
encode a synthetic 832x480 Main10 lowdelay_P clip with HM's encoder, then time
  (1) HM's own decoder (oracle/_ref/TAppDecoder, -O3, one thread) on the stream: whole-decoder wall time, and
  (2) the C restatement (oracle/hm_oracle.c) on the SAME pictures' parsed data (reconstruction + deblocking + SAO only),
so that the oracle-based `cpu_baseline` bench.py reports on the GPU box can be related to "HM single thread".
Writes profiles/cpu_calibration.json.  Not part of the test suite (takes minutes)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import hmref, hmoracle, make_golden      # noqa: E402
from libhm_amd import abi                            # noqa: E402
from tests import golden_util as gu                  # noqa: E402

W, H, FRAMES, BD, QP = 832, 480, 5, 10, 30


def main():
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "c.yuv")
        make_golden.write_yuv(yuv, make_golden.synth_clip(W, H, FRAMES, BD, seed=77, novel=True), BD)
        bs, rec = os.path.join(tmp, "c.bin"), os.path.join(tmp, "c_rec.yuv")
        cmd = [hmref.ENCODER_PATH, "-c", os.path.join(make_golden.HM_CFG, "encoder_lowdelay_P_main10.cfg"), "-i", yuv, "-wdt", str(W),
               "-hgt", str(H), "-fr", "30", "-f", str(FRAMES), "--InputBitDepth=%d" % BD, "--InternalBitDepth=%d" % BD, "-q", str(QP),
               "-b", bs, "-o", rec, "--SEIDecodedPictureHash=1", "--SearchRange=16", "--ECU=1", "--CFM=1", "--ESD=1"]
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        t_enc = time.time() - t0
        # (1) HM's decoder, best of 3
        dec = os.path.join(os.path.dirname(hmref.ENCODER_PATH), "TAppDecoder")
        t_hm = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            subprocess.run([dec, "-b", bs, "-o", os.path.join(tmp, "d.yuv")], check=True, stdout=subprocess.DEVNULL)
            t_hm = min(t_hm, time.perf_counter() - t0)
        # (2) the restatement on the parsed pictures (fixture dump in memory), best of 3 per picture
        data = open(bs, "rb").read()
        make_golden.GOLD = tmp
        make_golden.dump_stream("calib", data, np.fromfile(rec, dtype="<u2"), (W, H, FRAMES, BD))
        z = np.load(os.path.join(tmp, "stream_calib.npz"))
        poc_to_handle, pics = {}, []
        for i in range(int(z["num_pics"][0])):
            p = gu.Picture(z, i, poc_to_handle)
            poc_to_handle[p.poc] = i
            pics.append(p)
        hmoracle.lib()
        t_or, finals = 0.0, []
        for p in pics:
            best = 1e9
            for _ in range(3):
                cur = [np.zeros_like(a) for a in p.pre]
                t0 = time.perf_counter()
                hmoracle.decompress_ctus(p.seq, p.slices, p.meta, p.coeffs, cur, finals)
                hmoracle.loop_filter_pic(p.seq, p.slices, p.meta, p.pp, cur, 3)
                prm = hmoracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
                fin = hmoracle.sao_process(p.seq, p.slices, p.pp, p.meta, prm, cur)
                best = min(best, time.perf_counter() - t0)
            assert all(np.array_equal(a, b) for a, b in zip(fin, p.fin))
            t_or += best
            finals.append(p.fin)
    px = W * H * FRAMES
    out = {"clip": "%dx%d Main10 lowdelay_P, %d frames (1 I + %d P), QP %d, synthetic" % (W, H, FRAMES, FRAMES - 1, QP),
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip(), "threads": 1,
           "hm_decoder_s": round(t_hm, 4), "hm_decoder_Mpx_s": round(px / t_hm / 1e6, 2),
           "hm_decoder_note": "whole TAppDecoder process: parsing + reconstruction + loop filters + MD5 check + YUV write",
           "oracle_s": round(t_or, 4), "oracle_Mpx_s": round(px / t_or / 1e6, 2),
           "oracle_note": "hm_oracle.c: reconstruction (inter + intra) + deblocking + SAO of the same pictures from parsed data, bit-exact",
           "oracle_over_hm_time_ratio": round(t_or / t_hm, 3), "encode_s": round(t_enc, 1)}
    path = os.path.join(os.path.dirname(HERE), "profiles", "cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
