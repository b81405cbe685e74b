"""experiment: do two independent batches on two HIP streams (two contexts) overlap usefully on one GPU?
prints ms per batch of 8 pictures for 1 context and for 2 concurrent contexts"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import libhm_amd
from libhm_amd import abi
from tests import synth

w, h, bd, nb = 3840, 2160, 10, int(os.environ.get("NB", "8"))
metas = [synth.make_picture(w, h, bd, seed=0x484D3136 + i, bi=False, ref_handles=([0], [1])) for i in range(2)]
ref_planes = [synth.noise_planes(w, h, bd, 100), synth.blocky_planes(w, h, bd, 200)]


def make_ctx():
    ctx = libhm_amd.Context(abi.make_seq(w, h, bd, bd, log2_ctu=6, max_pictures=3 * nb), device=0)
    pics = []
    for i in range(nb):
        r0, r1 = ctx.acquire(), ctx.acquire()
        ctx.upload(r0, ref_planes[0]); ctx.upload(r1, ref_planes[1])
        hc = ctx.acquire()
        p = metas[i % 2]
        p.slice.ref_pic[0][0] = r0
        ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
        ctx.filter_picture(hc, p.pp, p.sao_raw)
        pics.append(hc)
    ctx.sync()
    return ctx, pics


N = 40
ctxs = [make_ctx() for _ in range(int(os.environ.get("NCTX", "2")))]
for c, p in ctxs:
    c.replay(p, 15, 3)
t0 = time.perf_counter(); ctxs[0][0].replay(ctxs[0][1], 15, N); ctxs[0][0].sync(); t1 = time.perf_counter() - t0
print("1 context : %.4f ms per batch of %d" % (t1 / N * 1e3, nb))
ths = [threading.Thread(target=lambda c=c, p=p: (c.replay(p, 15, N), c.sync())) for c, p in ctxs]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t2 = time.perf_counter() - t0
print("%d contexts: %.4f ms per batch of %d (wall %.2f ms for %d batches)" % (len(ctxs), t2 / (N * len(ctxs)) * 1e3, nb, t2 * 1e3, N * len(ctxs)))
