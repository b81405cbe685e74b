"""where a decoded picture's wall time goes: HMDEC_STATS=1 python tools/time_decoder.py <stream> [passes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libhm_amd import hmdec
data = open(sys.argv[1], "rb").read()
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nals = hmdec.split_nal_units(data)
t_push = t_get = 0.0
n = 0
with hmdec.Decoder(check_hash=False) as d:
    for k in range(passes):
        for i, nal in enumerate(nals):
            eof = k == passes - 1 and i == len(nals) - 1
            while True:
                t0 = time.perf_counter(); new_pic, check = d.push(nal, eof); t_push += time.perf_counter() - t0
                if check:
                    t0 = time.perf_counter()
                    while d.get_picture() is not None:
                        n += 1
                    t_get += time.perf_counter() - t0
                if not new_pic:
                    break
print("pictures %d: push %.1f ms (parse + device submission), get_picture %.1f ms (wait for the device + download)" % (n, t_push * 1e3, t_get * 1e3))
