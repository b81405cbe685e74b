import sys, numpy as np
sys.path.insert(0, '.')
import libhm_amd
from tests import kat_pictures as kp, golden_util as gu
z = gu.load("kats")
for bd in (8, 10):
    for is_chroma in (0, 1):
        for bi in (0, 1):
            p, ref, checks = kp.build(bd, is_chroma, bi)
            with libhm_amd.Context(p.seq) as ctx:
                h0, h1, hc = ctx.acquire(), ctx.acquire(), ctx.acquire()
                ctx.upload(h0, ref); ctx.upload(h1, ref); ctx.upload(hc, [np.zeros_like(a) for a in ref])
                ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs)
                got = ctx.download(hc)
            bad = []
            for comp, y, x, want, ka, kb in checks:
                g = got[comp][y:y + want.shape[0], x:x + want.shape[1]]
                if not np.array_equal(g, want): bad.append((comp, y, x, ka, kb, int((g != want).sum())))
            print("bd", bd, "chroma", is_chroma, "bi", bi, "bad", len(bad), "of", len(checks))
            cases = z["interp_cases_bd%d" % bd]
            for b in bad[:6]:
                print("   ", b, tuple(int(v) for v in cases[b[3]]))
            if bad:
                comp, y, x, ka, kb, n = bad[0]
                want = [c for c in checks if c[0] == comp and c[1] == y and c[2] == x][0][3]
                print(got[comp][y:y + want.shape[0], x:x + want.shape[1]]); print(want)
