"""Frame-parallel GOP path on the device (BASELINE config #5) with one rank: the level-batched replay of a
random-access GOP of 8 B pictures must give, picture by picture, the oracle's samples when the oracle decodes the same
pictures one after the other; and a finished picture moved through hmgpu_picture_device_region (the region an RCCL
send/recv would carry) must serve as a reference exactly like the original."""
import numpy as np
import pytest

from libhm_amd import abi, frame_parallel as fp
from tests import synth

pytestmark = pytest.mark.gpu


def _oracle_picture(oracle, p, sl, refs):
    cur = [np.zeros_like(a) for a in refs[0]]
    oracle.decompress_ctus(p.seq, [sl], p.meta, p.coeffs, cur, refs)
    oracle.loop_filter_pic(p.seq, [sl], p.meta, p.pp, cur, 3)
    prm = oracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
    return oracle.sao_process(p.seq, [sl], p.pp, p.meta, prm, cur)


@pytest.mark.parametrize("w,h", [(416, 240), (3840, 2160)])     # BASELINE config #5 at its full size too (one GOP, one GPU)
def test_gop_of_8_matches_oracle(oracle, w, h):
    import libhm_amd
    from libhm_amd import abi
    bd = 10
    pics = {poc: synth.make_picture(w, h, bd, seed=40 + poc, bi=True, ref_handles=([0], [0])) for poc in fp.RA_GOP8}
    anchor = synth.noise_planes(w, h, bd, 7)
    # oracle: decode order, every picture from the finished planes of its two references
    want = {0: anchor}
    for poc, (a, b) in fp.RA_GOP8.items():
        sl = abi.clone_slice(pics[poc].slice)
        sl.ref_pic[0][0], sl.ref_pic[1][0] = 0, 1
        want[poc] = _oracle_picture(oracle, pics[poc], sl, [want[a], want[b]])
    with libhm_amd.Context(abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=9)) as ctx:
        run = fp.DeviceGops(ctx, None, 0, 1, 1, lambda g, poc: pics[poc], lambda g: anchor)
        assert [len(lvl["compute"][0]) for lvl in run.plan] == [1, 1, 2, 4]
        for _ in range(2):                        # the replayed step must reproduce the staged run
            run.step()
        for poc in fp.RA_GOP8:
            got = ctx.download(run.handle_of[(0, poc)])
            for c in range(3):
                assert np.array_equal(got[c], want[poc][c]), "POC %d comp %d" % (poc, c)


def test_received_picture_is_a_reference_like_the_original():
    import torch
    import libhm_amd
    w, h, bd = 416, 240, 10
    a = synth.make_picture(w, h, bd, seed=3, ref_handles=([0], [0]))
    b = synth.make_picture(w, h, bd, seed=4, ref_handles=([0], [0]), mv_range=64)
    from libhm_amd import abi
    with libhm_amd.Context(abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=5)) as ctx:
        h0, ha, hb, hr, hc = (ctx.acquire() for _ in range(5))
        ctx.upload(h0, synth.noise_planes(w, h, bd, 1))
        ctx.decompress_slice(ha, 0, a.slice, a.meta, a.coeffs)
        ctx.filter_picture(ha, a.pp, a.sao_raw)
        # picture B predicted from the original A
        b.slice.ref_pic[0][0] = ha
        ctx.decompress_slice(hb, 0, b.slice, b.meta, b.coeffs)
        want = ctx.download(hb)
        # ... and from a copy of A that travelled region-to-region on the context's stream
        with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_handle())):
            src = fp.region_tensor(ctx, ha)
            dst = fp.region_tensor(ctx, hr, receive=True)
            assert src.numel() == dst.numel() and src.data_ptr() != dst.data_ptr()
            dst.copy_(src)
        ctx.commit_received(hr)
        for x, y in zip(ctx.download(hr), ctx.download(ha)):
            assert np.array_equal(x, y)
        b.slice.ref_pic[0][0] = hr
        ctx.decompress_slice(hc, 0, b.slice, b.meta, b.coeffs)
        got = ctx.download(hc)
        for x, y in zip(got, want):
            assert np.array_equal(x, y)


def _run_two_ranks(env_extra, nproc=2):
    """tools/gop_two_ranks.py under torch.distributed.run (fresh processes): every rank checks the pictures it owns against the oracle"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "gop_two_ranks.py")]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("0 mismatches") == nproc, r.stdout[-3000:]


def test_two_ranks_exchange_reference_pictures_on_one_gpu():
    """the multi-rank device path (plan, device regions, send/recv on the context's stream, commit) with two processes sharing the
    card and gloo carrying the regions: what a one-GPU box can run of BASELINE config #5"""
    _run_two_ranks({"HMGPU_DIST_BACKEND": "gloo", "HMGPU_SINGLE_DEVICE": "1"})


def test_two_ranks_exchange_reference_pictures_over_rccl():
    """the same with one GPU per rank and RCCL (needs two GPUs: skipped on the one-GPU boxes)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_two_ranks({})


def test_picture_transfer_between_contexts(oracle):
    """hmgpu_picture_transfer: a finished picture of one context becomes a reference picture of another (two contexts of one process --
    on different GPUs a peer copy over xGMI, here on the same one): the second context predicts from it and must get the oracle's picture"""
    import libhm_amd
    w, h, bd = 832, 480, 10
    p1 = synth.make_picture(w, h, bd, seed=71, intra_frac=0.1, ref_handles=([0], [0]))
    p2 = synth.make_picture(w, h, bd, seed=72, ref_handles=([0], [0]))
    ref = synth.noise_planes(w, h, bd, 73)
    cur = synth.blocky_planes(w, h, bd, 74)

    def chain(p, cur, refs):
        rec = [a.copy() for a in cur]
        oracle.decompress_ctus(p.seq, [p.slice], p.meta, p.coeffs, rec, refs)
        oracle.loop_filter_pic(p.seq, [p.slice], p.meta, p.pp, rec, 3)
        prm = oracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
        return oracle.sao_process(p.seq, [p.slice], p.pp, p.meta, prm, rec)
    want1 = chain(p1, cur, [ref])
    want2 = chain(p2, cur, [want1])
    with libhm_amd.Context(p1.seq) as a, libhm_amd.Context(p1.seq) as b:
        ha0, ha1 = a.acquire(), a.acquire()
        a.upload(ha0, ref)
        a.upload(ha1, cur)
        p1.slice.ref_pic[0][0] = ha0
        a.decompress_slice(ha1, 0, p1.slice, p1.meta, p1.coeffs)
        a.filter_picture(ha1, p1.pp, p1.sao_raw)
        hb_junk, hb0, hb1 = b.acquire(), b.acquire(), b.acquire()
        a.transfer_to(ha1, b, hb0)                       # no sync in between: the copy is ordered behind a's filters
        assert a.transfer_bytes > 2 * w * h
        b.upload(hb1, cur)
        p2.slice.ref_pic[0][0] = hb0
        b.decompress_slice(hb1, 0, p2.slice, p2.meta, p2.coeffs)
        b.filter_picture(hb1, p2.pp, p2.sao_raw)
        got1, got2 = b.download(hb0), b.download(hb1)
        for c in range(3):
            assert np.array_equal(got1[c], want1[c]), "transferred picture comp %d" % c
            assert np.array_equal(got2[c], want2[c]), "picture predicted from it, comp %d" % c
        with pytest.raises(libhm_amd.HmgpuError):
            a.transfer_to(ha1, a, ha1)
