#!/usr/bin/env python3
"""bench.py -- throughput of the HM pixel-reconstruction hot path on MI355X.

Workload (BASELINE.json: "decoded Mpixels/s + achieved HBM GB/s, 2160p Main10"): synthetic 3840x2160 Main10
lowdelay_P-shaped pictures (SURVEY.md 8d configs #3 + #4 in one: random CTU partitioning, one random quarter-sample MV
per PU, "typical" coefficients with cbf probability 0.5, QP 22..37, SAO parameters per CTU), pre-parsed, i.e. exactly
what HM's CABAC stage leaves behind.  One "step" = the whole device path -- prep (flattening) + luma/chroma motion
compensation + de-quantisation/inverse transform/reconstruction + deblocking (both edge directions) + SAO -- over one
batch of `--batch` independent pictures (frame-parallel, one launch per kernel for the batch), replayed from inputs that
are already resident in HBM (hmgpu_replay_batch).  The pictures use distinct device buffers (~110 MB each), so the
working set is far beyond the 256 MiB Infinity Cache.

`--workload` selects the SURVEY.md 8(d) sub-benchmarks instead (each prints the same JSON shape, `roofline` then names
that configuration's kernel):  idct = config #2 (1920x1080, every CTU four 32x32 luma + 16x16 chroma TUs, "stress" levels),
mc / mc_bi = config #3 (prep + MC + residual add only), filter = config #4 (25 % intra CUs for Bs = 2 edges; the
reconstruction stages run as well because they regenerate the pre-filter picture the in-place deblocking consumes).

Output: ONE JSON line (rank 0), see README/DESIGN.md.  `roofline` is PINNED to `mc_luma`, the kernel BASELINE.json's north star
puts a number on (sub-benchmarks name their own kernel) -- not the kernel with the largest share of device time, which is
`filter_fused`; `kernels` lists every kernel, `kernels_bi` the two motion-compensation kernels on B pictures.  `cpu_baseline` times
the oracle (oracle/hm_oracle.c, the C restatement pinned against HM) on this box's host cores on a bounded sample of the same
workload; `cpu_baseline_reference` is HM's own decoder (oracle/_ref/TAppDecoder, when the build container shipped it) on an
HM-encoded 2160p Main10 stream, HM's own per-picture decode time -- reported baselines, not targets.

`--gpus N` (N > 1) without a torch.distributed environment starts the N ranks itself (`python -m torch.distributed.run`, as a child
process, before anything touches the GPU) and relays rank 0's line; under an external launcher WORLD_SIZE must equal N.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes(p):
    """algorithmic HBM bytes per picture and kernel (DESIGN.md 'Algorithmic bytes'): int16 samples/levels, each counted once"""
    m = p.meta_np
    decoded = p.inside & (m["part_size"] != 8)
    inter = decoded & (m["pred_mode"] == 0)
    lists = (m["ref_idx0"] >= 0).astype(np.int64) + (m["ref_idx1"] >= 0).astype(np.int64)
    luma_inter = int(inter.sum()) * 16
    ref_reads = int((lists * inter).sum()) * 16            # luma samples read from reference pictures (one per list used)
    out = {}
    out["mc_luma"] = 2 * ref_reads + 2 * luma_inter
    out["mc_chroma"] = (2 * ref_reads + 2 * luma_inter) // 2
    # coded TU samples per size class: 2 (level) + 2 (residual written by k_itx) and 2 (residual read where MC writes the prediction)
    tr = m["tr_idx"]
    log2tu = 6 - m["depth"] - tr
    chain = (1 << (tr + 1)) - 1
    per_cls = {2: 0, 3: 0, 4: 0, 5: 0}
    coded = [0, 0, 0]
    n_tu = 0                                               # transform units k_prep lists (12-byte records)
    for comp, key in enumerate(("cbf_y", "cbf_u", "cbf_v")):
        has = inter & ((m[key] & chain) == chain)
        for l2 in (2, 3, 4, 5):
            n_part = int((has & (log2tu == l2)).sum())     # partitions covered by coded TUs whose luma node is 2^l2
            if comp == 0:
                per_cls[l2] += n_part * 16
            else:
                cls = max(l2 - 1, 2)
                per_cls[cls] += n_part * 4
            coded[comp] += n_part * (16 if comp == 0 else 4)
            # partitions per TU node: (2^(l2-2))^2; the one 4x4 chroma TU of four 4x4 luma TUs counts once per 8x8 area
            n_tu += n_part // ((1 << (2 * (l2 - 2))) if (comp == 0 or l2 > 2) else 4)
    out["itx"] = 4 * sum(per_cls.values())                # all four TU sizes run in one launch
    out["mc_luma"] += 2 * coded[0]
    out["mc_chroma"] += 2 * (coded[1] + coded[2])
    samples = p.width * p.height * 3 // 2
    out["deblock_ver"] = 2 * samples                       # each pass = half of the 4 B/sample two-pass budget
    out["deblock_hor"] = 2 * samples
    out["sao"] = 4 * samples
    out["filter_fused"] = 4 * samples                      # deblocking (both directions) + SAO in one pass: picture read once, written once
    # k_prep: 21 B of HM's arrays read per partition (part size, depth, pred mode, QP, transform index, bypass, PCM, two reference
    # indices, two vectors, three cbf bytes), per 8x8 area (four partitions) one 16-byte TileMv and one 8-byte EdgeRec written, 12 B per
    # listed transform unit (the 16-byte BlkInfo records are only stored for calls whose kernels read them: not this workload)
    out["prep"] = int(decoded.sum()) * (21 + 4 + 2) + 12 * n_tu
    # intra CUs: levels read (coded TUs), reconstruction written once, reference samples read (~ 4N+1 per N x N TU: counted as 2 B/sample)
    intra_p = decoded & (m["pred_mode"] == 1)
    out["intra"] = int(intra_p.sum()) * 24 * (2 + 2 + 2)
    # replicated margins written around the three final planes (128/64 samples left+right, 80/40 rows above+below)
    out["extend_border"] = 2 * ((2 * 128 * (p.height + 160) + 160 * p.width) + 2 * (2 * 64 * (p.height // 2 + 80) + 80 * (p.width // 2)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="independent pictures per step (<= 16)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bi", type=int, default=0, help="1: B pictures (bi-prediction) instead of P")
    ap.add_argument("--workload", default="full", choices=("full", "idct", "mc", "mc_bi", "filter", "intra", "gop", "decode"))
    ap.add_argument("--stream", default=None, help="decode workload: Annex B file (default tests/golden/bench_ldp_wpp_main10_3840x2160.bin: 2160p with wavefronts)")
    ap.add_argument("--mode-probs", default=None, help="experiment: CTU partition probabilities 64x64,32x32,16x16,8x8,AMP (comma separated)")
    ap.add_argument("--intra-frac", type=float, default=None, help="experiment: fraction of CUs that are intra (with intra modes: reconstructed on the GPU)")
    ap.add_argument("--cbf-prob", type=float, default=None, help="experiment: probability that a TU is coded")
    ap.add_argument("--mv-range", type=int, default=None, help="experiment: integer MV range in luma samples (default 64)")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2), help="2: the batch runs as two half-batches on two HIP streams (kernels of different kinds overlap)")
    ap.add_argument("--threads", type=int, default=8, help="decode workload: parser threads of libhmdec (1 = all on the calling thread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the host-inclusive (staging included) measurement of the default workload")
    ap.add_argument("--profile-steps", type=int, default=5)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args.gpus)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    import torch  # noqa: F401  -- first, so that libhmgpu.so binds to the HIP runtime torch ships (one runtime per process)
    from libhm_amd import dist as hdist
    dist, rank, world, local_rank = hdist.init_from_env()
    if world > 1:
        # every rank says where it runs: a scaling run is self-evidencing
        sys.stderr.write("bench rank %d of %d: backend %s (world size %d), device cuda:%d of %d visible\n" % (
            rank, world, dist.get_backend(), dist.get_world_size(), local_rank, torch.cuda.device_count()))
    if os.environ.get("HMGPU_BENCH_RENDEZVOUS_ONLY"):
        # rehearsal of the launch path (tests/test_distributed_cpu.py, no GPU needed): the ranks meet, agree on the slowest one's clock
        # through the same barrier + MAX as the timed region, rank 0 prints the line's launch-related fields, nothing is measured
        el = hdist.timed_region(dist, lambda: None, lambda: None)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rendezvous_only": True, "backend": dist.get_backend() if dist is not None else None,
                              "elapsed_max_over_ranks_s": el}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    copy_gbps = measured_copy_bandwidth(local_rank) if world == 1 else None

    import libhm_amd
    from libhm_amd import abi
    from tests import synth

    if args.workload == "gop":
        return gop_main(args, hdist, dist, rank, world, local_rank, copy_gbps)
    if args.workload == "decode":
        return decode_main(args, hdist, dist, rank, world, local_rank)
    w, h, bd = args.width, args.height, 10
    nb = args.batch
    wl = args.workload
    RECON, FILTER = 8, 7
    # the roofline object is pinned to the motion-compensation kernel, the one kernel BASELINE.json's north star puts a number on
    # (>= 40 % of the HBM roofline); `kernels` lists all of them
    stages, roof_kernel, kw = RECON | FILTER, "mc_luma", {}
    if wl == "idct":
        w, h = 1920, 1080
        stages, roof_kernel = RECON, "itx"
        kw = dict(mode_probs=(1.0, 0, 0, 0, 0), cbf_prob=1.0, coef_dist="stress", sao=False)
    if wl in ("mc", "mc_bi"):
        stages, roof_kernel = RECON, "mc_luma"
        args.bi = 1 if wl == "mc_bi" else 0
    if wl == "filter":
        # the reconstruction stages run too: they regenerate the pre-filter picture that the in-place deblocking consumed
        roof_kernel = "sao"
        kw = dict(intra_frac=0.25, intra_modes=False)      # the intra CUs only supply Bs = 2 edges; their reconstruction is the "intra" workload
    if wl == "intra":
        # all-intra pictures (I pictures): the serial chain of the path, a CTU-row wavefront on the device
        roof_kernel = "intra"
        kw = dict(intra_frac=1.0)
    if args.cbf_prob is not None:
        kw["cbf_prob"] = args.cbf_prob
    if args.intra_frac is not None:
        kw["intra_frac"] = args.intra_frac
    if args.mv_range is not None:
        kw["mv_range"] = args.mv_range
    if args.mode_probs:
        kw["mode_probs"] = tuple(float(v) for v in args.mode_probs.split(","))
    # two distinct parsed pictures, staged alternately into nb device pictures with their own buffers
    # every picture of the batch predicts from its OWN reference pictures (no flattering reuse of one reference in cache)
    metas = [synth.make_picture(w, h, bd, seed=0x484D3136 + 7 * rank + i, bi=bool(args.bi), ref_handles=([0], [1]), **kw) for i in range(2)]
    host_incl = wl == "full" and world == 1 and not args.no_host_inclusive and 4 * nb <= 64
    seq = abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=(4 if host_incl else 3) * nb)
    ctx = libhm_amd.Context(seq, device=local_rank)
    ref_planes = [synth.noise_planes(w, h, bd, 100 + rank), synth.blocky_planes(w, h, bd, 200 + rank)]
    pics, refs_of = [], []
    t_stage = time.time()
    for i in range(nb):
        r0, r1 = ctx.acquire(), ctx.acquire()
        refs_of.append((r0, r1))
        ctx.upload(r0, ref_planes[0])
        ctx.upload(r1, ref_planes[1])
        hc = ctx.acquire()
        p = metas[i % 2]
        for l, r in ((0, r0), (1, r1)):
            if p.slice.num_ref_idx[l] > 0:
                p.slice.ref_pic[l][0] = r
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)      # stages the inputs in HBM (and runs once)
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        pics.append(hc)
    ctx.sync()
    t_stage = time.time() - t_stage

    ALL = stages
    ctx.set_streams(args.streams)
    for _ in range(args.warmup):
        ctx.replay(pics, ALL, 1)
    # barrier + sync on both sides, MAX over ranks (libhm_amd/dist.py)
    elapsed = hdist.timed_region(dist, lambda: ctx.replay(pics, ALL, args.steps), ctx.sync,
                                 device=("cuda:%d" % local_rank) if dist is not None else None)

    # the same steps on ONE stream (kernel after kernel: what the per-kernel times below add up to)
    ms_one_stream = None
    if args.streams == 2 and world == 1:
        ctx.set_streams(1)
        ctx.replay(pics, ALL, 1)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.replay(pics, ALL, args.steps)
        ctx.sync()
        ms_one_stream = (time.perf_counter() - t1) / args.steps * 1e3
        ctx.set_streams(args.streams)
    # ---- per-kernel device times: hipEvents on the context's own stream, around every launch of extra steps (one stream while profiling)
    ctx.set_profiling(True)
    ctx.stats(reset=True)
    ctx.replay(pics, ALL, args.profile_steps)
    st = ctx.stats(reset=True)
    ctx.set_profiling(False)

    # ---- host-inclusive steady state (SURVEY 8d "slice" figure): the same pictures through the product entry points
    # hmgpu_decompress_pictures + hmgpu_filter_pictures, every input travelling from page-locked staging blocks over PCIe each
    # step; two sets of device pictures, so that the inputs of one step are copied while the kernels of the previous one run
    hi = None
    if host_incl:
        hi = host_inclusive(ctx, metas, pics, refs_of, nb, w, h, args)

    if rank == 0:
        luma_px = w * h
        total_px = world * nb * args.steps * luma_px
        value = total_px / elapsed / 1e6
        bytes_pp = [algorithmic_bytes(m) for m in metas]
        kernels = {}
        dom, dom_t = None, -1.0
        for name, (ms, launches) in st["kernels"].items():
            if launches == 0 or name in ("h2d_stage", "other", ""):
                continue
            avg_ms = ms / launches
            per_launch = sum(bytes_pp[i % 2][name] for i in range(nb))
            gbs = per_launch / (avg_ms * 1e-3) / 1e9
            kernels[name] = {"avg_ms": round(avg_ms, 5), "alg_MB": round(per_launch / 1e6, 2), "GBps": round(gbs, 1),
                             "frac": round(gbs / HBM_PEAK_GBS, 4)}
            if ms > dom_t:
                dom, dom_t = name, ms
        if roof_kernel == "sao" and "sao" not in kernels:
            roof_kernel = "filter_fused"                   # all three loop-filter stages run as one kernel when every picture has SAO
        if roof_kernel is not None:
            dom = roof_kernel
        # HBM traffic per launch: counters cannot be read from inside the process, so the figure comes from the PMC passes of this
        # same command (tools/round_profile.sh + tools/pmc_summary.py -> profiles/hbm_traffic.json) -- and only when that file was
        # taken from the kernel sources this run is built from (sha256 over libhm_amd/csrc), else null
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if wl == "full" and nb == 16 and not args.bi and (w, h) == (3840, 2160) and os.path.exists(tpath):
            tj = json.load(open(tpath))
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pmc_summary_digest
            if tj.get("csrc_sha16") == pmc_summary_digest.csrc_digest():
                traffic = tj["kernels"].get(dom, {}).get("traffic_bytes")
                traffic_source = "profiles/hbm_traffic.json@csrc:" + tj["csrc_sha16"]
            else:
                traffic_source = "none: profiles/hbm_traffic.json was taken from other kernel sources (%s)" % tj.get("csrc_sha16")
        roof = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["frac"], "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(kernels[dom]["alg_MB"] * 1e6)}
        dev_ms = sum(k["avg_ms"] for k in kernels.values())
        out = {
            "metric": "decoded Mpixels/s (luma), 2160p Main10 reconstruction + loop filters" if wl == "full" else
                      "Mpixels/s (luma) through the '%s' stage set (SURVEY 8d sub-benchmark)" % wl, "value": round(value, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 samples / int32 accumulate", "data": "synthetic",
            "config": {"workload": "%dx%d Main10 %s pictures, pre-parsed CTU metadata: %s, "
                                   "batch of %d independent pictures per step, inputs resident in HBM" %
                                   (w, h, "B (bi-pred)" if args.bi else "lowdelay_P",
                                    {"full": "prep+MC+dequant/IT/recon+deblock+SAO", "idct": "prep+MC+dequant/IT/recon, 32x32 luma / 16x16 chroma TUs, stress levels",
                                     "mc": "prep+MC+dequant/IT/recon", "mc_bi": "prep+MC+dequant/IT/recon",
                                     "intra": "all-intra pictures: prep+intra prediction/dequant/IT/recon (CTU-row wavefront)+deblock+SAO", "filter": "prep+MC+dequant/IT/recon+deblock+SAO, 25% intra CUs (Bs 2 edges)"}[wl], nb),
                       "sub_benchmark": wl, "streams": args.streams,
                       "pictures_per_step": nb, "parallelism": "frame-parallel, 1 process per GPU, no data-path collective"},
            "roofline": roof, "kernels": kernels, "device_ms_per_step_sum_of_kernels": round(dev_ms, 4),
            "hbm_GBps_whole_step_algorithmic": round(sum(sum(b[k] for k in kernels) for b in [bytes_pp[i % 2] for i in range(nb)])
                                                    / (elapsed / args.steps) / 1e9, 1),
            "staging_s_for_batch_incl_first_run": round(t_stage, 3),
        }
        if ms_one_stream is not None:
            out["ms_per_step_one_stream"] = round(ms_one_stream, 4)
        if copy_gbps is not None:
            out["hbm_copy_GBps_measured"] = copy_gbps
        if hi is not None:
            out.update(hi)
        if wl == "full" and world == 1 and not args.bi:
            # the two motion-compensation kernels on B pictures (config #3's bi-pred variant, 6 B per sample), beside the P mix above
            ctx.close()
            ctx = None
            out["kernels_bi"] = mc_bi_kernels(w, h, bd, nb, rank, local_rank, args)
        if not args.no_cpu_baseline and world == 1 and wl == "full":
            out["cpu_baseline"] = cpu_baseline(metas[0], w, h, bd, 1, 10.0)
            out["cpu_baseline_all_cores"] = cpu_baseline(metas[0], w, h, bd, 0, 10.0)
            ref = cpu_baseline_reference()
            if ref is not None:
                out["cpu_baseline_reference"] = ref
                out["speedup_vs_HM_1thread"] = round(value / ref["value"], 1)
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def mc_bi_kernels(w, h, bd, nb, rank, local_rank, args):
    """`--workload mc_bi` in short, for the default line: the same mix as B pictures (70 % of the PUs bi-predicted), prep + MC + residual
    only, per-kernel hipEvent times of the two motion-compensation kernels against their algorithmic bytes"""
    import libhm_amd
    from libhm_amd import abi
    from tests import synth
    metas = [synth.make_picture(w, h, bd, seed=0x484D3136 + 7 * rank + i, bi=True, ref_handles=([0], [1])) for i in range(2)]
    ctx = libhm_amd.Context(abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=3 * nb), device=local_rank)
    ref_planes = [synth.noise_planes(w, h, bd, 100 + rank), synth.blocky_planes(w, h, bd, 200 + rank)]
    pics = []
    for i in range(nb):
        r0, r1 = ctx.acquire(), ctx.acquire()
        ctx.upload(r0, ref_planes[0])
        ctx.upload(r1, ref_planes[1])
        hc = ctx.acquire()
        p = metas[i % 2]
        for l, r in ((0, r0), (1, r1)):
            if p.slice.num_ref_idx[l] > 0:
                p.slice.ref_pic[l][0] = r
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        pics.append(hc)
    ctx.sync()
    ctx.set_streams(1)
    for _ in range(args.warmup):
        ctx.replay(pics, 8, 1)
    ctx.set_profiling(True)
    ctx.stats(reset=True)
    ctx.replay(pics, 8, max(args.profile_steps, 5))
    st = ctx.stats(reset=True)
    ctx.set_profiling(False)
    ctx.close()
    bytes_pp = [algorithmic_bytes(m) for m in metas]
    out = {}
    for name in ("mc_luma", "mc_chroma"):
        ms, launches = st["kernels"][name]
        avg_ms = ms / launches
        per_launch = sum(bytes_pp[i % 2][name] for i in range(nb))
        gbs = per_launch / (avg_ms * 1e-3) / 1e9
        out[name] = {"avg_ms": round(avg_ms, 5), "alg_MB": round(per_launch / 1e6, 2), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def launch_ranks(n):
    """`bench.py --gpus N` called directly: start the N ranks as ONE child process tree (torch.distributed.run, one rank per GPU,
    rendezvous over 127.0.0.1) before this process has touched the GPU, pass their output through (rank 0 prints the JSON line)
    and exit with the child's status.  Nothing is exec'ed: the parent only waits."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.stderr.write("bench.py: a rank failed (exit status %d)\n" % rc)
    sys.exit(rc if rc >= 0 else 1)


def host_inclusive(ctx, metas, pics, refs_of, nb, w, h, args):
    """steady-state throughput of the product path with host -> device staging inside the timed region"""
    from libhm_amd import abi
    import libhm_amd
    stg, level_bytes = [], 0
    for i in range(nb):                                        # one staging block per picture of a step (page-locked, filled once)
        s = ctx.staging_alloc()
        # levels in the compact form (coded TUs only, hmgpu_pack_levels): what a parser that appends TU after TU produces
        level_bytes = s.fill_compact(libhm_amd.lib(), ctx.seq, metas[i % 2].meta, metas[i % 2].coeffs)
        mnp = metas[i % 2].meta_np
        has_intra = bool((mnp["pred_mode"] == 1).any())
        s.set_groups(intra=has_intra, flags=False)           # (the synthetic pictures use no transform skip / lossless / PCM CUs)
        stg.append(s)
    sets = [pics, [ctx.acquire() for _ in range(nb)]]
    slices = []
    for i in range(nb):
        sl = abi.clone_slice(metas[i % 2].slice)
        for l, r in ((0, refs_of[i][0]), (1, refs_of[i][1])):
            if sl.num_ref_idx[l] > 0:
                sl.ref_pic[l][0] = r
        slices.append(sl)
    sao = [abi.sao_array_from_raw(m.sao_raw) for m in metas]
    djobs = [ctx.picture_jobs([(hset[i], [slices[i]], stg[i], stg[i]) for i in range(nb)]) for hset in sets]
    fjobs = [ctx.filter_jobs([(hset[i], metas[i % 2].pp, sao[i % 2]) for i in range(nb)]) for hset in sets]

    def step(k):
        ctx.decompress_pictures(djobs[k & 1])
        ctx.filter_pictures(fjobs[k & 1])
    for k in range(4):
        step(k)
    ctx.sync()
    n = max(4, args.steps)
    t0 = time.perf_counter()
    for k in range(n):
        step(k)
    t_issue = time.perf_counter() - t0                        # the host's share: the calls return once everything is enqueued
    ctx.sync()
    dt = time.perf_counter() - t0
    a0 = stg[0].arrays
    group = {"base": ["slice_idx", "tile_idx", "depth", "part_size", "pred_mode", "qp", "tr_idx", "cbf_y", "cbf_u", "cbf_v", "mv0", "ref_idx0"],
             "list1": ["mv1", "ref_idx1"], "intra": ["intra_dir_l", "intra_dir_c"]}
    meta_bytes = sum(a0[k].nbytes for k in group["base"]) + (sum(a0[k].nbytes for k in group["list1"]) if args.bi else 0) + \
        (sum(a0[k].nbytes for k in group["intra"]) if has_intra else 0)
    staged = meta_bytes + level_bytes + 3 * 4 * (ctx.num_ctus + 1)
    dense = sum(a.nbytes for a in stg[0].arrays.values()) + sum(a.nbytes for a in stg[0].levels)
    for s in stg:
        ctx.staging_free(s)
    return {"host_inclusive_Mpixels_s": round(n * nb * w * h / dt / 1e6, 1),
            "host_inclusive": {"ms_per_step": round(dt / n * 1e3, 3), "steps": n, "staged_bytes_per_picture": int(staged),
                               "staged_bytes_per_picture_dense_levels": int(dense),
                               "PCIe_GBps": round(n * nb * staged / dt / 1e9, 1), "host_issue_ms_per_step": round(t_issue / n * 1e3, 3),
                               "what": "hmgpu_decompress_pictures + hmgpu_filter_pictures per step, inputs copied from page-locked staging "
                                       "blocks every step (the metadata a P picture without intra CUs needs in one DMA, compact levels -- coded TUs only -- in "
                                       "three, on a copy stream; two sets of device pictures: "
                                       "the copies of a step overlap the kernels of the previous one)"}}


def gop_main(args, hdist, dist, rank, world, local_rank, copy_gbps):
    """BASELINE config #5: random-access GOPs of 8 B pictures, frame-parallel over the ranks, `world` GOPs in flight; finished
    reference pictures travel rank-to-rank over RCCL send/recv.  value = pictures of all ranks per second (8 per rank, step)."""
    import libhm_amd
    from libhm_amd import abi, frame_parallel as fp
    from tests import synth
    w, h, bd = args.width, args.height, 10
    gops = world
    metas = [synth.make_picture(w, h, bd, seed=0x484D3136 + 5 + i, bi=True, ref_handles=([0], [0])) for i in range(2)]
    ctx = libhm_amd.Context(abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=9 * gops), device=local_rank)
    anchors = [synth.noise_planes(w, h, bd, 100), synth.blocky_planes(w, h, bd, 200)]
    t0 = time.time()
    run = fp.DeviceGops(ctx, dist, rank, world, gops, lambda g, poc: metas[(g + poc) % 2], lambda g: anchors[g % 2])
    t_stage = time.time() - t0
    for _ in range(args.warmup):
        run.step()

    def steps():
        for _ in range(args.steps):
            run.step()
    elapsed = hdist.timed_region(dist, steps, ctx.sync, device=("cuda:%d" % local_rank) if dist is not None else None)
    ctx.set_profiling(True)
    ctx.stats(reset=True)
    run.step()
    st = ctx.stats(reset=True)
    ctx.set_profiling(False)
    # every rank says what it moved: a scaling run is self-evidencing (which backend carried the pictures, how many ranks, how many bytes)
    my_sends = sum(len(s[3]) for lvl in run.plan for s in lvl["sends"] if s[2] == rank)
    my_recvs = sum(1 for lvl in run.plan for s in lvl["sends"] if rank in s[3])
    sys.stderr.write("gop rank %d of %d: backend %s, %d pictures sent / %d received per step, %d bytes each (device region to device region)\n" % (
        rank, world, (dist.get_backend() if dist is not None else "none (one rank)"), my_sends, my_recvs, ctx.device_region(run.handle_of[(0, 0)])[1]))
    if rank == 0:
        plan = run.plan
        sends = sum(len(s[3]) for lvl in plan for s in lvl["sends"])
        region = ctx.device_region(run.handle_of[(0, 0)])[1]
        out = {
            "metric": "decoded Mpixels/s (luma), 2160p Main10 random-access GOPs, frame-parallel", "unit": "Mpixels/s",
            "value": round(world * 8 * args.steps * w * h / elapsed / 1e6, 1), "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 samples / int32 accumulate", "data": "synthetic",
            "config": {"workload": "%dx%d Main10 random-access GOP of 8 B pictures (POC 8<-0; 4<-0,8; 2<-0,4; 6<-4,8; odd<-neighbours), "
                                   "%d GOPs in flight, pictures of one dependency level batched per rank, inputs resident in HBM" % (w, h, gops),
                       "sub_benchmark": "gop", "pictures_per_step": 8 * world, "dependency_levels": len(plan),
                       "parallelism": "frame-parallel, 1 process per GPU, RCCL send/recv of finished reference pictures",
                       "transfers_per_step": sends, "bytes_per_transfer": region},
            "kernels_rank0_ms_per_step": {k: round(ms, 4) for k, (ms, n) in st["kernels"].items() if n},
            "staging_s": round(t_stage, 3),
        }
        if copy_gbps is not None:
            out["hbm_copy_GBps_measured"] = copy_gbps
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def decode_main(args, hdist, dist, rank, world, local_rank):
    """SURVEY 8 f-2 sub-benchmark: a whole decoder.  An HM-encoded Annex B stream goes through libhmdec (host parser + device
    reconstruction) exactly as a libHM client would drive it; a step is one pass over the stream, every rank decodes its own copy
    (replicas).  Verified by the stream's own decoded-picture-hash SEI in an untimed pass; the timed passes run with the check off
    (as the HM baseline does) but hand every picture to the "application" (device -> host planes), as libHMDec_get_picture does."""
    import subprocess
    from libhm_amd import hmdec
    here = os.path.dirname(os.path.abspath(__file__))
    path = args.stream or os.path.join(here, "tests", "golden", "bench_ldp_wpp_main10_3840x2160.bin")
    data = open(path, "rb").read()
    nals = hmdec.split_nal_units(data)
    info = {}

    def drain(d, state):
        while True:
            p = d.get_picture()
            if p is None:
                return
            w, h = p.size(0)
            state["pixels"] += w * h
            state["pictures"] += 1
            info["size"] = (w, h)

    def feed(d, state, last):
        """one pass of the stream through an existing decoder (libHM's loop); the next pass's IDR flushes what is still held"""
        for i, nal in enumerate(nals):
            eof = last and i == len(nals) - 1
            while True:
                new_pic, check = d.push(nal, eof)
                if check:
                    drain(d, state)
                if not new_pic:
                    break

    # verification pass (MD5 of every picture against the SEI of the stream)
    state = {"pixels": 0, "pictures": 0}
    with hmdec.Decoder(device=local_rank, check_hash=True, threads=args.threads) as d:
        feed(d, state, True)
        npic, pixels, bad = d.pictures_decoded, state["pixels"], d.hash_mismatches
    if bad or npic == 0 or state["pictures"] != npic:
        raise SystemExit("decode: %d of %d pictures disagree with their hash SEI (%d output)" % (bad, npic, state["pictures"]))
    # timed: ONE decoder (device context, picture buffers) fed the stream args.steps times, as a player looping a clip would --
    # with the decoded-picture-hash SEI check ON, libHM's default (the MD5s run on the decoder's hash threads), and once more with
    # the check off (what HM's own timing below does)
    def timed(check, device_md5=False):
        d = hmdec.Decoder(device=local_rank, check_hash=check, threads=args.threads, device_md5=device_md5)
        state = {"pixels": 0, "pictures": 0}
        for _ in range(max(1, args.warmup)):
            feed(d, state, False)
        state["pixels"] = state["pictures"] = 0

        def steps():
            for k in range(args.steps):
                feed(d, state, k == args.steps - 1)
        t = hdist.timed_region(dist, steps, lambda: None, device=("cuda:%d" % local_rank) if dist is not None else None)
        t_tail = time.perf_counter()
        bad_now = d.hash_mismatches                          # (waits for the checks still under way: the tail of the last pictures' chains)
        t_tail = time.perf_counter() - t_tail
        batches = d.device_batches
        d.close()
        timed.tail_s, timed.batches = t_tail, batches
        if bad_now:
            raise SystemExit("decode: %d pictures disagree with their hash SEI in the timed passes" % bad_now)
        # the warm-up passes leave pictures in the decoder that the first timed pass puts out: count what actually came out
        return t, state["pixels"], state["pictures"]
    elapsed_off, pixels_off, pictures_off = timed(False)
    # check on: the MD5 chains run on the device (hmgpu_picture_hash_begin: no download for the check, no hash threads; libhmdec's default).
    # The chains of the last pictures finish after the last picture is out -- a GPU lane runs an MD5 chain ~8x slower than a host core --
    # that tail is reported separately.  Beside it: the check on the decoder's hash threads (round 2's form).
    elapsed, pixels_timed, pictures_timed = timed(True, device_md5=True)
    batches_timed, tail_dev = timed.batches, timed.tail_s
    elapsed_host, _, pictures_host = timed(True, device_md5=False)
    # host parsing alone (no device work): what bounds the decoder today
    t0 = time.perf_counter()
    with hmdec.Decoder(parse_only=True, threads=args.threads) as d:
        d.decode_stream(data)
    t_parse = time.perf_counter() - t0
    if rank == 0:
        w, h = info["size"]
        out = {
            "metric": "decoded Mpixels/s (luma), whole decoder: Annex B stream -> pictures at the application", "unit": "Mpixels/s",
            "value": round(world * pixels_timed / elapsed / 1e6, 1), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 samples / int32 accumulate", "data": "synthetic clip encoded by HM 16.0 (%s, %d bytes, %d pictures)" % (os.path.basename(path), len(data), npic),
            "config": {"workload": "%dx%d HM-encoded stream through libhmdec: host CABAC parsing (%d parser thread%s) + device reconstruction + "
                                   "picture download, %d NAL units" % (w, h, args.threads, "" if args.threads == 1 else "s, frame-parallel", len(nals)),
                       "sub_benchmark": "decode", "parser_threads": args.threads, "pictures_per_step": npic, "parallelism": "replicas, 1 process per GPU"},
            "fps": round(world * pictures_timed / elapsed, 2),
            "hash_sei_check": "on in the timed passes (MD5 of every picture, chains on the device: one lane per plane, batches of up to 32 pictures)",
            "fps_hash_check_off": round(world * pictures_off / elapsed_off, 2),
            "fps_hash_check_on_host_threads": round(world * pictures_host / elapsed_host, 2),
            "device_md5_tail_s": round(tail_dev, 3),
            "device_batches_per_step": round(batches_timed / max(1, args.steps + max(1, args.warmup)), 2),
            "host_parse_only_Mpixels_s": round(pixels / t_parse / 1e6, 1),
            "host_parse_only_Mbit_s": round(len(data) * 8 / t_parse / 1e6, 1),
            "hash_sei_verified_pictures": npic,
            "roofline": None,        # host-bound (CABAC parsing); the device kernels are measured by the default workload
        }
        ref_dec = os.path.join(here, "oracle", "_ref", "TAppDecoder")
        if not args.no_cpu_baseline and os.path.exists(ref_dec):
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                subprocess.run([ref_dec, "-b", path, "--SEIDecodedPictureHash=0"], check=True, stdout=subprocess.DEVNULL)
                best = min(best, time.perf_counter() - t0)
            out["cpu_baseline"] = {"value": round(pixels / best / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                                   "sample": "HM 16.0 TAppDecoder (oracle/_ref, -O3) on the same stream, hash check off, no output file, best of 2: %.2f s" % best}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def measured_copy_bandwidth(device):
    """device-to-device copy of 1 GiB (read + write counted), the practical HBM ceiling on this box"""
    import torch
    dev = "cuda:%d" % device
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    gbps = round(10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b
    torch.cuda.empty_cache()
    return gbps


def cpu_baseline_reference(stream="bench_ldp_main10_3840x2160.bin"):
    """HM itself beside the headline: oracle/_ref/TAppDecoder (HM 16.0 compiled from the reference sources in the build container;
    the binary travels to the GPU box, the sources do not) on an HM-encoded 3840x2160 Main10 lowdelay_P stream, hash check off, no
    output file.  The figure is HM's own per-picture decode time -- the [DT] column of its log, clock() around decompressSlice +
    filterPicture (TDecGop.cpp:113,154,162,186-188), CABAC parsing included, as HM does not separate it -- summed over the pictures;
    the whole process beside it.  None when the binary or the stream is not there."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppDecoder")
    path = os.path.join(ROOT, "tests", "golden", stream)
    if not (os.path.exists(exe) and os.path.exists(path)):
        return None
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([exe, "-b", path, "--SEIDecodedPictureHash=0"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        except (subprocess.SubprocessError, OSError):
            return None
        wall = time.perf_counter() - t0
        dts = [float(m.group(1)) for m in re.finditer(r"\[DT\s+([0-9.]+)\]", r.stdout)]
        types = re.findall(r"\(\s*([IPB])-SLICE", r.stdout)
        if not dts:
            return None
        if best is None or sum(dts) < best[0]:
            best = (sum(dts), dts, types, wall)
    dt, dts, types, wall = best
    m = re.search(r"_(\d+)x(\d+)", stream)
    w, h = (int(m.group(1)), int(m.group(2))) if m else (3840, 2160)
    px = len(dts) * w * h
    p_dts = [d for d, t in zip(dts, types) if t == "P"]
    model = cpu_model()
    out = {"value": round(px / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference", "cpu": model,
           "sample": "HM 16.0 TAppDecoder (oracle/_ref, g++ -O3) on tests/golden/%s: %d pictures %dx%d (%s), hash check off, no output file; "
                     "sum of HM's own [DT] per picture %.3f s (parsing + reconstruction + loop filters), best of 2; whole process %.2f s" %
                     (stream, len(dts), w, h, "".join(types), dt, wall),
           "whole_process_Mpixels_s": round(px / wall / 1e6, 2)}
    if p_dts:
        out["P_pictures_Mpixels_s"] = round(len(p_dts) * w * h / sum(p_dts) / 1e6, 2)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(p, w, h, bd, threads, seconds):
    """the oracle (C restatement of HM, pinned against HM goldens) on this box's host: whole pictures of the same workload.
    threads = 1: one picture after the other on one core; threads = 0: one worker per core this process may use, every worker
    its own independent pictures (frame-parallel, as the device batch is) -- the C calls release the interpreter lock"""
    import threading
    from oracle import hmoracle
    from tests import synth
    from libhm_amd import abi
    hmoracle.lib()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                   # a container's CPU share (cgroup v2), when it is smaller than the affinity mask
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = min(cores, 32)                                 # (bounded: every worker holds a 2160p picture and its references)
    nthr = cores if threads == 0 else threads
    refs = [synth.noise_planes(w, h, bd, 100), synth.blocky_planes(w, h, bd, 200)]
    sl = abi.clone_slice(p.slice)             # the device run re-pointed the reference handles at its own pictures
    for l in range(2):
        if sl.num_ref_idx[l] > 0:
            sl.ref_pic[l][0] = l
    done = [0] * nthr
    t_end = [0.0]

    def one_picture():
        cur = [np.zeros_like(r) for r in refs[0]]
        hmoracle.decompress_ctus(p.seq, [sl], p.meta, p.coeffs, cur, refs)
        hmoracle.loop_filter_pic(p.seq, [sl], p.meta, p.pp, cur, 3)
        prm = hmoracle.sao_reconstruct_params(p.seq, p.pp, p.meta, p.sao_raw)
        hmoracle.sao_process(p.seq, [sl], p.pp, p.meta, prm, cur)

    def worker(i, deadline):
        while time.perf_counter() < deadline and done[i] < 40:
            one_picture()
            done[i] += 1
        t_end[0] = max(t_end[0], time.perf_counter())

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(i, t0 + seconds)) for i in range(nthr)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    t_total = t_end[0] - t0
    n = sum(done)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n * w * h / t_total / 1e6, 2), "unit": "Mpixels/s", "cores": nthr, "kind": "port", "cpu": model,
            "sample": "%d whole %dx%d pictures of the same synthetic workload through oracle/hm_oracle.c (recon+deblock+SAO) on %d thread%s, %.1f s" %
                      (n, w, h, nthr, "" if nthr == 1 else "s (one picture each at a time)", t_total)}


if __name__ == "__main__":
    main()
