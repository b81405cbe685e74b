import os, sys
sys.path.insert(0, os.getcwd())
os.environ["HMDEC_PLACE_ROUND_ROBIN"] = "1"
os.environ["HMDEC_STATS"] = "1"
from libhm_amd import hmdec
from tests import golden_util as gu
z = gu.load("stream_ra_main10_208x120")
with hmdec.Decoder(threads=1, devices=[0, 0]) as d:
    d.decode_stream(z["bitstream"])
    print("decoded", d.pictures_decoded, "batches", d.device_batches, "devices", d.num_devices, "moved", d.transfer_bytes, "mismatch", d.hash_mismatches)
