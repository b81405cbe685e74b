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
