import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import libhm_amd
from libhm_amd import abi
from tests import synth
w, h, bd = 3840, 2160, 10
p = synth.make_picture(w, h, bd, seed=1, ref_handles=([0], [0]))
ctx = libhm_amd.Context(p.seq)
r = ctx.acquire(); ctx.upload(r, synth.noise_planes(w, h, bd, 3))
hc = ctx.acquire()
ctx.decompress_slice(hc, 0, p.slice, p.meta, p.coeffs); ctx.sync()
out = (C.c_double * 4)()
L = libhm_amd.lib()
L.hmgpu_debug_mc_stamps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
for i in range(3):
    st = L.hmgpu_debug_mc_stamps(ctx._h, hc, out)
    print("status", st, "meta-wait %.0f cyc, window+compute+store %.0f cyc, waves %d, launch span %.0f cyc" % (out[0], out[1], out[2], out[3]))
