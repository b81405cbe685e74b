#!/usr/bin/env python3
"""Frame-parallel GOP parity with several ranks: every rank runs its share of `world` random-access GOPs through
libhm_amd.frame_parallel.DeviceGops (finished reference pictures travel between the ranks' device picture regions) and checks the
pictures it owns against the oracle, which decodes the same GOPs serially.  Launch under torchrun, e.g. on a one-GPU box
  HMGPU_DIST_BACKEND=gloo HMGPU_SINGLE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/gop_two_ranks.py
(gloo moves the regions through the host; with one GPU per rank the default backend is RCCL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    from libhm_amd import abi, dist as hdist, frame_parallel as fp
    import libhm_amd
    from oracle import hmoracle
    from tests import synth
    dist, rank, world, local_rank = hdist.init_from_env()
    w, h, bd = 416, 240, 10
    gops = world
    pics = {(g, poc): synth.make_picture(w, h, bd, seed=1000 + 16 * g + poc, bi=True, intra_frac=0.05, ref_handles=([0], [0]))
            for g in range(gops) for poc in fp.RA_GOP8}
    anchors = {g: synth.noise_planes(w, h, bd, 50 + g) for g in range(gops)}
    hmoracle.lib()
    want = {}
    for g in range(gops):
        fin = {0: anchors[g]}
        for poc, (a, b) in fp.RA_GOP8.items():
            p = pics[(g, poc)]
            sl = abi.clone_slice(p.slice)
            sl.ref_pic[0][0], sl.ref_pic[1][0] = 0, 1
            cur = [np.zeros_like(x) for x in anchors[g]]
            hmoracle.decompress_ctus(p.seq, [sl], p.meta, p.coeffs, cur, [fin[a], fin[b]])
            hmoracle.loop_filter_pic(p.seq, [sl], p.meta, p.pp, cur, 3)
            prm = hmoracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
            fin[poc] = hmoracle.sao_process(p.seq, [sl], p.pp, p.meta, prm, cur)
            want[(g, poc)] = fin[poc]
    dev = 0 if os.environ.get("HMGPU_SINGLE_DEVICE") else local_rank
    bad = 0
    with libhm_amd.Context(abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=9 * gops), device=dev) as ctx:
        run = fp.DeviceGops(ctx, dist, rank, world, gops, lambda g, poc: pics[(g, poc)], lambda g: anchors[g])
        for _ in range(2):
            run.step()
        for key in run.mine:
            got = ctx.download(run.handle_of[key])
            if not all(np.array_equal(a, b) for a, b in zip(got, want[key])):
                bad += 1
                print("rank %d: GOP %d POC %d differs from the oracle" % (rank, key[0], key[1]), file=sys.stderr)
        print("rank %d of %d: %d pictures owned, %d mismatches, %d transfers per step in the plan" %
              (rank, world, len(run.mine), bad, sum(len(s[3]) for lvl in run.plan for s in lvl["sends"])), file=sys.stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
