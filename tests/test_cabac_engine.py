"""libhmdec's arithmetic decoding engine (branch-free bins, grouped bypass bins)
against a literal transcription of Rec. ITU-T H.265 9.3.4.3 on random data: same bins, same context states, same bit positions.
HM counterpart: TDecBinCABAC (TDecBinCoderCABAC.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_equals_the_specifications_procedure(tmp_path):
    exe = str(tmp_path / "cabac_engine_test")
    dec = os.path.join(ROOT, "libhm_amd", "dec")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", dec, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "cabac_engine_test.cpp"), os.path.join(dec, "cabac.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout
