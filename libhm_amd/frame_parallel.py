"""Frame-parallel reconstruction of one GOP over several GPUs (BASELINE.json config #5).

A picture depends on other pictures only through the finished planes of its reference pictures (TComPrediction.cpp:593).
The plan below assigns every picture to a rank (pictures of one dependency level round-robin) and lists, in decode order,
which finished pictures have to be delivered to which ranks.  Delivery is ONE collective per referenced picture: a
broadcast of its planes from the owner (xGMI is point-to-point, a one-to-many fan-out uses all links of the owner at once;
a ring would be bound by a single link).  Pictures nobody references are never sent.
"""


def plan_gop(refs_of, world):
    """refs_of: dict poc -> list of reference pocs, in DECODE order (insertion order).  Returns
    (owner: poc -> rank, level: poc -> dependency depth, sends: list of (poc, src_rank, sorted dst ranks) in decode order)."""
    level, owner, per_level = {}, {}, {}
    for poc, refs in refs_of.items():
        for r in refs:
            if r not in level:
                raise ValueError("reference %d of picture %d is not decoded before it" % (r, poc))
        level[poc] = 1 + max((level[r] for r in refs), default=-1)
        k = per_level.get(level[poc], 0)
        owner[poc] = k % world
        per_level[level[poc]] = k + 1
    sends = []
    for poc in refs_of:
        dst = sorted({owner[p] for p, refs in refs_of.items() if poc in refs and owner[p] != owner[poc]})
        if dst:
            sends.append((poc, owner[poc], dst))
    return owner, level, sends


def critical_path(refs_of):
    """number of dependency levels = lower bound of sequential steps for the GOP"""
    _, level, _ = plan_gop(refs_of, 1)
    return 1 + max(level.values())


def run_gop(refs_of, rank, world, reconstruct, alloc, broadcast):
    """Generic executor (the same code drives device pictures over RCCL and CPU tensors over gloo in the tests).
    reconstruct(poc, {ref poc: buffer}) -> buffer     called on the owner
    alloc(poc) -> empty buffer                         called on ranks that receive a picture
    broadcast(buffer, src_rank)                        collective, called by EVERY rank for every sent picture
    Returns {poc: buffer} of everything this rank owns or received."""
    owner, _, sends = plan_gop(refs_of, world)
    send_set = {poc: (src, dst) for poc, src, dst in sends}
    have = {}
    for poc, refs in refs_of.items():
        if owner[poc] == rank:
            have[poc] = reconstruct(poc, {r: have[r] for r in refs})
        if poc in send_set:
            src, dst = send_set[poc]
            if rank != src and rank in dst:
                have[poc] = alloc(poc)
            # every rank takes part in the collective; ranks that do not need the picture pass a scratch buffer
            buf = have[poc] if (rank == src or rank in dst) else alloc(poc)
            broadcast(buf, src)
    return have


# ---------------------------------------------------------------------------------------------------------------------
# Pipelined GOPs: `gops` independent GOPs (closed GOPs / separate streams) are in flight at once and the pictures of one
# dependency level -- of ALL GOPs -- form one batch per rank, so every rank has work at every level even though a single
# random-access GOP of 8 only offers 1, 1, 2 and 4 independent pictures ("with >= 4 GOPs in flight
# the ideal is 8x").  Finished pictures go point-to-point to exactly the ranks that predict from them (xGMI is a
# point-to-point fabric: one send per needing rank uses that rank's own link, nobody else's).

# random-access GOP of 8 (cfg/encoder_randomaccess_main10.cfg:22-31): POC -> (list-0 reference, list-1 reference), decode order
RA_GOP8 = {8: (0, 0), 4: (0, 8), 2: (0, 4), 6: (4, 8), 1: (0, 2), 3: (2, 4), 5: (4, 6), 7: (6, 8)}


def plan_pipelined(refs_of, world, gops, anchor=0):
    """refs_of: poc -> reference pocs in decode order; `anchor` is available on every rank beforehand (previous GOP).
    Picture j (decode index) of GOP g belongs to rank (g + j) % world.
    Returns a list over dependency levels of {"compute": {rank: [(g, poc)]}, "sends": [(g, poc, src, [dst...])]}"""
    order = list(refs_of)
    level = {anchor: -1}
    for poc in order:
        for r in refs_of[poc]:
            if r not in level:
                raise ValueError("reference %d of picture %d is not decoded before it" % (r, poc))
        level[poc] = 1 + max(level[r] for r in refs_of[poc])
    owner = {(g, poc): (g + j) % world for g in range(gops) for j, poc in enumerate(order)}
    plan = [{"compute": {}, "sends": []} for _ in range(1 + max(level[p] for p in order))]
    for g in range(gops):
        for poc in order:
            plan[level[poc]]["compute"].setdefault(owner[(g, poc)], []).append((g, poc))
            dst = sorted({owner[(g, q)] for q in order if poc in refs_of[q] and owner[(g, q)] != owner[(g, poc)]})
            if dst:
                plan[level[poc]]["sends"].append((g, poc, owner[(g, poc)], dst))
    return plan


def run_pipelined(plan, rank, reconstruct_batch, exchange):
    """reconstruct_batch([(g, poc), ...]): this rank's pictures of one level (mutually independent);
    exchange([(g, poc, src, [dst...]), ...]): the transfers of that level this rank takes part in, in plan order"""
    for lvl in plan:
        mine = lvl["compute"].get(rank, [])
        if mine:
            reconstruct_batch(mine)
        part = [s for s in lvl["sends"] if s[2] == rank or rank in s[3]]
        if part:
            exchange(part)


class _DeviceBytes:
    """a device address range as something torch.as_tensor understands (no copy, no ownership)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def region_tensor(ctx, pic, receive=False):
    """uint8 tensor aliasing the plane region of a device picture (hmgpu_picture_device_region)"""
    import torch
    ptr, nbytes = ctx.device_region(pic, receive)
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda:%d" % ctx.device)


def exchange_device(dist, ctx, rank, handle_of, transfers):
    """RCCL send/recv of finished pictures between the plane regions of the ranks' contexts, ordered on the context's stream"""
    import torch
    ops, received = [], []
    with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_handle(), device="cuda:%d" % ctx.device)):
        for g, poc, src, dst in transfers:
            pic = handle_of[(g, poc)]
            if rank == src:
                t = region_tensor(ctx, pic)
                ops += [dist.P2POp(dist.isend, t, d) for d in dst]
            else:
                ops.append(dist.P2POp(dist.irecv, region_tensor(ctx, pic, receive=True), src))
                received.append(pic)
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for pic in received:
        ctx.commit_received(pic)


class DeviceGops:
    """`gops` random-access GOPs in flight over `world` ranks, on device pictures of one hmgpu context per rank.
    picture_of(g, poc) -> object with .slice/.meta/.coeffs/.pp/.sao_raw (parsed picture); anchor_planes(g) -> the three planes
    of POC 0 (previous GOP's last picture, resident everywhere).  Every picture is staged once (inputs into HBM); step()
    then replays reconstruction + loop filters level by level and ships finished reference pictures between the ranks."""

    def __init__(self, ctx, dist, rank, world, gops, picture_of, anchor_planes, refs_of=None):
        self.ctx, self.dist, self.rank, self.world = ctx, dist, rank, world
        self.refs_of = refs_of or RA_GOP8
        self.plan = plan_pipelined(self.refs_of, world, gops)
        self.handle_of = {}
        for g in range(gops):
            self.handle_of[(g, 0)] = ctx.acquire()
            ctx.upload(self.handle_of[(g, 0)], anchor_planes(g))
            for poc in self.refs_of:
                self.handle_of[(g, poc)] = ctx.acquire()
        self.mine = []
        for lvl in self.plan:
            for g, poc in lvl["compute"].get(rank, []):
                p = picture_of(g, poc)
                a, b = self.refs_of[poc]
                p.slice.ref_pic[0][0] = self.handle_of[(g, a)]
                if p.slice.num_ref_idx[1] > 0:
                    p.slice.ref_pic[1][0] = self.handle_of[(g, b)]
                h = self.handle_of[(g, poc)]
                ctx.decompress_slice(h, 0, p.slice, p.meta, p.coeffs)       # copies the inputs to the device (and runs once)
                ctx.filter_picture(h, p.pp, p.sao_raw)
                self.mine.append((g, poc))
        ctx.sync()

    def step(self):
        run_pipelined(self.plan, self.rank,
                      lambda items: self.ctx.replay([self.handle_of[k] for k in items], 15, 1),
                      lambda tr: exchange_device(self.dist, self.ctx, self.rank, self.handle_of, tr))
