"""The N > 1 path on CPU: two ranks over gloo (127.0.0.1) exercise the sharding, the barrier / MAX-over-ranks timing contract
of bench.py, and the frame-parallel GOP plan with its broadcast of finished reference pictures."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from libhm_amd import dist as hdist
from libhm_amd import frame_parallel as fp

# random-access GOP-8 of cfg/encoder_randomaccess_main10.cfg: POC 8 <- 0; 4 <- 0,8; 2 <- 0,4; 6 <- 4,8; odd <- neighbours
RA_GOP8 = {0: [], 8: [0], 4: [0, 8], 2: [0, 4], 6: [4, 8], 1: [0, 2], 3: [2, 4], 5: [4, 6], 7: [6, 8]}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_reconstruct(poc, refs):
    out = torch.full((4, 6), float(poc + 1))
    for r in sorted(refs):
        out = out * 1.0 + refs[r] * 0.5
    return out


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist, r, w, _ = hdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # 1. sharding of independent units
    b, e = hdist.shard(11, w, r)
    # 2. timing contract: rank 1 is slower, every rank must see its time
    import time
    el = hdist.timed_region(dist, lambda: time.sleep(0.05 * (r + 1)), lambda: None)
    # 3. frame-parallel GOP with broadcasts of reference pictures
    have = fp.run_gop(RA_GOP8, r, w, _fake_reconstruct, lambda poc: torch.zeros(4, 6),
                      lambda buf, src: dist.broadcast(buf, src=src))
    owner, _, _ = fp.plan_gop(RA_GOP8, w)
    mine = {p: have[p].numpy().copy() for p in RA_GOP8 if owner[p] == r}
    q.put((r, (b, e), el, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    # shards tile [0, 11) exactly
    assert res[0][1] == (0, 6) and res[1][1] == (6, 11)
    # MAX over ranks: both ranks report (about) the slow rank's time
    assert abs(res[0][2] - res[1][2]) < 1e-6 and res[0][2] >= 0.1
    # the distributed GOP equals the serial one
    serial = {}
    for poc, refs in RA_GOP8.items():
        serial[poc] = _fake_reconstruct(poc, {r: serial[r] for r in refs})
    got = {}
    for r in res:
        got.update(r[3])
    assert sorted(got) == sorted(RA_GOP8)
    for poc in RA_GOP8:
        assert np.array_equal(got[poc], serial[poc].numpy())


def _fake_recon2(g, poc, r0, r1):
    return torch.full((4, 6), float(10 * g + poc + 1)) + 0.5 * r0 + 0.25 * r1


def _pipelined_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist, r, w, _ = hdist.init_from_env(backend="gloo")
    gops = w
    plan = fp.plan_pipelined(fp.RA_GOP8, w, gops)
    have = {(g, 0): torch.full((4, 6), float(g)) for g in range(gops)}       # the anchors are everywhere

    def reconstruct_batch(items):
        for g, poc in items:
            a, b = fp.RA_GOP8[poc]
            have[(g, poc)] = _fake_recon2(g, poc, have[(g, a)], have[(g, b)])

    def exchange(transfers):
        ops = []
        for g, poc, src, dst in transfers:
            if r == src:
                ops += [dist.P2POp(dist.isend, have[(g, poc)], d) for d in dst]
            else:
                have[(g, poc)] = torch.zeros(4, 6)
                ops.append(dist.P2POp(dist.irecv, have[(g, poc)], src))
        for wk in dist.batch_isend_irecv(ops):
            wk.wait()

    fp.run_pipelined(plan, r, reconstruct_batch, exchange)
    mine = {k: have[k].numpy().copy() for lvl in plan for k in lvl["compute"].get(r, [])}
    q.put((r, mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_gops_over_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {}
    for _, mine in res:
        got.update(mine)
    assert len(got) == 8 * world
    for g in range(world):
        serial = {0: torch.full((4, 6), float(g))}
        for poc, (a, b) in fp.RA_GOP8.items():
            serial[poc] = _fake_recon2(g, poc, serial[a], serial[b])
            assert np.array_equal(got[(g, poc)], serial[poc].numpy())


def test_pipelined_plan_properties():
    for world in (1, 2, 4, 8):
        plan = fp.plan_pipelined(fp.RA_GOP8, world, world)
        assert len(plan) == 4                                    # critical path of the random-access GOP of 8
        per_rank = {}
        for lvl in plan:
            for r, items in lvl["compute"].items():
                per_rank[r] = per_rank.get(r, 0) + len(items)
            for g, poc, src, dst in lvl["sends"]:
                assert src not in dst and all(0 <= d < world for d in dst)
        assert per_rank == {r: 8 for r in range(world)}         # weak scaling: 8 pictures per rank and step
        if world == 1:
            assert all(not lvl["sends"] for lvl in plan)
        if world == 8:
            assert [len(lvl["compute"][0]) for lvl in plan] == [1, 1, 2, 4]


def test_gop_plan_properties():
    owner, level, sends = fp.plan_gop(RA_GOP8, 8)
    assert fp.critical_path(RA_GOP8) == 5                  # 0 | 8 | 4 | 2,6 | 1,3,5,7
    assert level == {0: 0, 8: 1, 4: 2, 2: 3, 6: 3, 1: 4, 3: 4, 5: 4, 7: 4}
    assert len({owner[p] for p in (1, 3, 5, 7)}) == 4      # one temporal level spreads over ranks
    sent = {p for p, _, _ in sends}
    assert sent <= {0, 8, 4, 2, 6} and not sent & {1, 3, 5, 7}   # only referenced pictures travel
    for poc, src, dst in sends:
        assert src == owner[poc] and src not in dst
    with pytest.raises(ValueError):
        fp.plan_gop({4: [0], 0: []}, 2)
    # one rank: nothing to send
    assert fp.plan_gop(RA_GOP8, 1)[2] == []


# ---- bench.py --gpus N called the way the driver's one-process form calls it: the script itself starts the ranks ------------------
def _run_bench(extra_args, env_extra, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HMGPU_BENCH_RENDEZVOUS_ONLY="1", HMGPU_DIST_BACKEND="gloo", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` without a torch.distributed environment: two ranks come up (torch.distributed.run as a child
    process), meet over gloo, every rank reports itself on stderr and rank 0 prints ONE line with n_gpus = 2."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "3"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["backend"] == "gloo"
    for rank in range(2):
        assert "bench rank %d of 2: backend gloo (world size 2)" % rank in r.stderr


def test_bench_gpus_flag_must_match_the_launcher():
    """under an external launcher WORLD_SIZE must equal --gpus: a scaling run cannot silently report another rank count"""
    r, lines = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, timeout=60)
    assert r.returncode != 0 and not lines and "WORLD_SIZE=3" in r.stderr
    r, lines = _run_bench(["--gpus", "1"], {})
    assert r.returncode == 0 and lines[0]["n_gpus"] == 1
