"""oracle/make_surgery.py -- fixtures for syntax HM's ENCODER never writes: PPS-level scaling lists, long-term reference pictures,
reference picture list modification (Rec. ITU-T H.265 7.3.2.3, 7.3.4, 7.3.6.1, 7.3.6.2).

TEST INFRASTRUCTURE ONLY (builder container: needs oracle/_ref, i.e. /root/reference).  An HM-encoded stream is rewritten at the bit level
-- parameter sets and slice segment headers get the extra syntax, the CABAC slice data is carried over byte for byte -- and HM's own
DECODER (oracle/_ref/TAppDecoder) decodes the result: its output pictures are the expected values of the fixture.  The pictures are
not what the encoder meant (other scaling factors, other reference pictures): what matters is that every decoder must produce exactly
these.  Usage: python oracle/make_surgery.py  ->  tests/golden/lite_surgery_*.npz (same layout as make_golden.make_lite).
"""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg   # noqa: E402

DECODER_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "TAppDecoder")


# ------------------------------------------------------------------------------------------------------------------ bits
class Bits:
    """reader over a string of '0' / '1'"""
    def __init__(self, s):
        self.s, self.p = s, 0

    def u(self, n):
        v = int(self.s[self.p:self.p + n], 2) if n else 0
        self.p += n
        return v

    def ue(self):
        z = 0
        while self.s[self.p] == "0":
            z += 1
            self.p += 1
        self.p += 1
        return (1 << z) - 1 + self.u(z)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


def w_u(v, n):
    return format(v, "0%db" % n) if n else ""


def w_ue(v):
    b = format(v + 1, "b")
    return "0" * (len(b) - 1) + b


def w_se(v):
    return w_ue(2 * v - 1 if v > 0 else -2 * v)


def split_nals(data):
    """Annex B byte stream -> list of NAL units (header + payload, emulation prevention still inside)"""
    out, i, n, start = [], 0, len(data), None
    while i + 3 <= n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            if start is not None:
                end = i
                while end > start and data[end - 1] == 0:
                    end -= 1
                out.append(bytes(data[start:end]))
            start = i + 3
            i += 3
        else:
            i += 1
    if start is not None:
        out.append(bytes(data[start:]))
    return out


def nal_bits(nal):
    """payload of a NAL unit (after the 2-byte header) as a bit string, emulation prevention bytes removed"""
    raw, z = bytearray(), 0
    for b in nal[2:]:
        if z >= 2 and b == 3:
            z = 0
            continue
        raw.append(b)
        z = z + 1 if b == 0 else 0
    return "".join(format(b, "08b") for b in raw)


def make_nal(header, bits):
    assert len(bits) % 8 == 0
    raw = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    out, z = bytearray(header), 0
    for b in raw:
        if z >= 2 and b <= 3:
            out.append(3)
            z = 0
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


def with_trailing(bits):
    """rbsp_trailing_bits: a 1 and zeros up to the byte boundary"""
    bits += "1"
    return bits + "0" * (-len(bits) % 8)


def strip_trailing(bits):
    return bits[:bits.rindex("1")]


# -------------------------------------------------------------------------------------------------------- parameter sets
def parse_st_rps(r, idx, sets, in_slice=False):
    """st_ref_pic_set(idx) (7.3.7 / 7.4.8): returns [(delta_poc, used_by_curr)] sorted as S0 then S1"""
    inter = r.u(1) if idx != 0 else 0
    if inter:
        delta_idx = r.ue() + 1 if in_slice else 1
        ref = sets[idx - delta_idx]
        sign, absd = r.u(1), r.ue() + 1
        drps = (1 - 2 * sign) * absd
        used, use_delta = [], []
        for _ in range(len(ref) + 1):
            ub = r.u(1)
            used.append(ub)
            use_delta.append(1 if ub else r.u(1))
        ref_neg = [e for e in ref if e[0] < 0]
        ref_pos = [e for e in ref if e[0] > 0]
        nn, npos = len(ref_neg), len(ref_pos)
        s0, s1 = [], []
        # (7-61)
        for j in range(npos - 1, -1, -1):
            d = ref_pos[j][0] + drps
            if d < 0 and use_delta[nn + j]:
                s0.append((d, used[nn + j]))
        if drps < 0 and use_delta[len(ref)]:
            s0.append((drps, used[len(ref)]))
        for j in range(nn):
            d = ref_neg[j][0] + drps
            if d < 0 and use_delta[j]:
                s0.append((d, used[j]))
        # (7-62)
        for j in range(nn - 1, -1, -1):
            d = ref_neg[j][0] + drps
            if d > 0 and use_delta[j]:
                s1.append((d, used[j]))
        if drps > 0 and use_delta[len(ref)]:
            s1.append((drps, used[len(ref)]))
        for j in range(npos):
            d = ref_pos[j][0] + drps
            if d > 0 and use_delta[nn + j]:
                s1.append((d, used[nn + j]))
        return s0 + s1
    nneg, npos = r.ue(), r.ue()
    out, d = [], 0
    for _ in range(nneg):
        d -= r.ue() + 1
        out.append((d, r.u(1)))
    d = 0
    for _ in range(npos):
        d += r.ue() + 1
        out.append((d, r.u(1)))
    return out


def parse_sps(bits):
    r = Bits(bits)
    s = {}
    r.u(4)
    msl = r.u(3)
    r.u(1)
    r.u(88 + 8)                                   # general profile / tier / level
    if msl:
        present = [(r.u(1), r.u(1)) for _ in range(msl)]
        r.u(2 * (8 - msl))
        for pp, lp in present:
            r.u(88 * pp + 8 * lp)
    r.ue()
    s["chroma"] = r.ue()
    assert s["chroma"] == 1
    s["w"], s["h"] = r.ue(), r.ue()
    if r.u(1):
        for _ in range(4):
            r.ue()
    r.ue(); r.ue()
    s["log2_poc"] = r.ue() + 4
    sub = r.u(1)
    for _ in range(0 if sub else msl, msl + 1):
        r.ue(); r.ue(); r.ue()
    log2_min_cb = r.ue() + 3
    s["log2_ctb"] = log2_min_cb + r.ue()
    r.ue(); r.ue(); r.ue(); r.ue()
    s["pos_scaling_list_enabled"] = r.p
    s["scaling_list_enabled"] = r.u(1)
    if s["scaling_list_enabled"]:
        assert r.u(1) == 0, "SPS carries scaling list data: not handled"
    r.u(1)
    s["sao"] = r.u(1)
    if r.u(1):
        r.u(8); r.ue(); r.ue(); r.u(1)
    nsets = r.ue()
    s["rps"] = []
    for i in range(nsets):
        s["rps"].append(parse_st_rps(r, i, s["rps"]))
    s["pos_lt_present"] = r.p
    s["lt_present"] = r.u(1)
    assert not s["lt_present"]
    s["tmvp"] = r.u(1)
    return s


def parse_pps(bits):
    r = Bits(bits)
    p = {}
    r.ue(); r.ue()
    p["dep_slices"] = r.u(1)
    p["output_flag_present"] = r.u(1)
    p["extra_bits"] = r.u(3)
    r.u(1)
    p["cabac_init_present"] = r.u(1)
    p["num_ref_l0"], p["num_ref_l1"] = r.ue() + 1, r.ue() + 1
    r.se()
    r.u(1); r.u(1)
    if r.u(1):
        r.ue()
    r.se(); r.se()
    p["slice_chroma_qp"] = r.u(1)
    p["wp"], p["wbp"] = r.u(1), r.u(1)
    r.u(1)
    p["tiles"], p["wpp"] = r.u(1), r.u(1)
    assert not p["tiles"] and not p["wpp"] and not p["wp"] and not p["wbp"]
    p["lf_across_slices"] = r.u(1)
    p["dbk_override_enabled"], p["dbk_disabled"] = 0, 0
    if r.u(1):
        p["dbk_override_enabled"] = r.u(1)
        p["dbk_disabled"] = r.u(1)
        if not p["dbk_disabled"]:
            r.se(); r.se()
    p["pos_scaling_list_present"] = r.p
    assert r.u(1) == 0
    p["pos_lists_modification"] = r.p
    p["lists_modification"] = r.u(1)
    r.ue()
    p["header_extension"] = r.u(1)
    assert not p["header_extension"]
    return p


# ----------------------------------------------------------------------------------------------------------- scaling lists
def random_scaling_list_data(rng):
    """scaling_list_data() (7.3.4) with every coding choice in use: predicted from the default list, copied from the previous matrix,
    explicit with and without DC"""
    out = ""
    for size_id in range(4):
        step = 3 if size_id == 3 else 1
        for matrix_id in range(0, 6, step):
            choice = rng.random()
            if choice < 0.15:
                out += "0" + w_ue(0)                                   # scaling_list_pred_matrix_id_delta 0: the default list
            elif choice < 0.35 and matrix_id > 0:
                out += "0" + w_ue(1)                                   # copy of the matrix before (its DC included)
            else:
                out += "1"
                nxt = 8
                if size_id > 1:
                    dc = rng.randint(4, 60)
                    out += w_se(dc - 8)
                    nxt = dc
                for _ in range(min(64, 1 << (4 + (size_id << 1)))):
                    target = min(250, max(2, nxt + rng.randint(-6, 9)))
                    delta = target - nxt
                    out += w_se(delta)
                    nxt = (nxt + delta + 256) % 256
    return out


# -------------------------------------------------------------------------------------------------------------- slice header
def rewrite_slice_header(bits, nal_type, sps, pps, state, lt_poc=None, modify=True):
    """slice_segment_header() of an independent P slice segment with long-term / list-modification syntax spliced in.  state: POC
    bookkeeping across calls.  Returns the new RBSP bit string."""
    r = Bits(bits)
    first = r.u(1)
    assert first, "one slice segment per picture expected"
    irap = 16 <= nal_type <= 23
    if irap:
        r.u(1)
    r.ue()
    r.u(pps["extra_bits"])
    slice_type = r.ue()
    if pps["output_flag_present"]:
        r.u(1)
    idr = nal_type in (19, 20)
    ins1 = ""
    poc = 0
    used_st = 0
    st_pocs = []
    pos1 = r.p
    if not idr:
        lsb = r.u(sps["log2_poc"])
        maxlsb = 1 << sps["log2_poc"]
        prev_lsb, prev_msb = state["prev_lsb"], state["prev_msb"]
        if lsb < prev_lsb and prev_lsb - lsb >= maxlsb // 2:
            msb = prev_msb + maxlsb
        elif lsb > prev_lsb and lsb - prev_lsb > maxlsb // 2:
            msb = prev_msb - maxlsb
        else:
            msb = prev_msb
        poc = msb + lsb
        if r.u(1):
            n = len(sps["rps"])
            idx = r.u(max(0, (n - 1).bit_length())) if n > 1 else 0
            rps = sps["rps"][idx]
        else:
            rps = parse_st_rps(r, len(sps["rps"]), sps["rps"], in_slice=True)
        used_st = sum(u for _, u in rps)
        st_pocs = [poc + d for d, _ in rps]
        pos1 = r.p
        # long-term part (the SPS now says long_term_ref_pics_present_flag = 1, num_long_term_ref_pics_sps = 0)
        use_lt = lt_poc is not None and poc > lt_poc + 1 and lt_poc not in st_pocs
        if use_lt:
            ins1 = w_ue(1) + w_u(lt_poc % maxlsb, sps["log2_poc"]) + "1" + "0"     # one picture, used by the current one, no MSB
        else:
            ins1 = w_ue(0)
        state["prev_lsb"], state["prev_msb"] = lsb, msb      # (TemporalId 0 throughout, no RASL / RADL / sub-layer non-reference pictures)
    else:
        state["prev_lsb"], state["prev_msb"] = 0, 0
        use_lt = False
    tmvp = 0
    if not idr and sps["tmvp"]:
        tmvp = r.u(1)
    sao_l = sao_c = 0
    if sps["sao"]:
        sao_l, sao_c = r.u(1), r.u(1)
    ins2, pos2 = "", r.p
    if slice_type != 2:
        assert slice_type == 1, "P slices expected"
        nref = pps["num_ref_l0"]
        if r.u(1):
            nref = r.ue() + 1
        pos2 = r.p
        total = used_st + (1 if use_lt else 0)
        if total > 1:
            if modify:
                # the temporary list is the short-term pictures in RPS order, then the long-term one: put the long-term picture first (or, without
                # one, turn the list around)
                v = (total - 1).bit_length()
                order = [total - 1] + list(range(total - 1)) if use_lt else list(range(total - 1, -1, -1))
                ins2 = "1" + "".join(w_u(order[i % total], v) for i in range(nref))
            else:
                ins2 = "0"
        if pps["cabac_init_present"]:
            r.u(1)
        if tmvp and nref > 1:
            r.ue()
        r.ue()                                                # five_minus_max_num_merge_cand
    r.se()                                                    # slice_qp_delta
    if pps["slice_chroma_qp"]:
        r.se(); r.se()
    dbk_disabled = pps["dbk_disabled"]
    if pps["dbk_override_enabled"]:
        if r.u(1):
            dbk_disabled = r.u(1)
            if not dbk_disabled:
                r.se(); r.se()
    if pps["lf_across_slices"] and (sao_l or sao_c or not dbk_disabled):
        r.u(1)
    end = r.p
    assert bits[end] == "1" and set(bits[end + 1:end + 1 + (-(end + 1) % 8)]) <= {"0"}, "byte_alignment() expected at the end of the header"
    data = end + 1 + (-(end + 1) % 8)
    head = bits[:pos1] + ins1 + bits[pos1:pos2] + ins2 + bits[pos2:end]
    head += "1"
    head += "0" * (-len(head) % 8)
    state["log"].append((poc, st_pocs, use_lt, ins2))
    return head + bits[data:]


# ------------------------------------------------------------------------------------------------------------------ drivers
def decode_with_hm(stream, w, h, frames, bd, tmp, name):
    bs = os.path.join(tmp, name + ".bin")
    yuv = os.path.join(tmp, name + ".yuv")
    with open(bs, "wb") as f:
        f.write(stream)
    r = subprocess.run([DECODER_PATH, "-b", bs, "-o", yuv, "-d", str(bd), "--SEIDecodedPictureHash=0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if r.returncode != 0:
        print(r.stdout[-3000:])
        raise RuntimeError("HM's decoder refused " + name)
    per = w * h * 3 // 2
    rec = np.fromfile(yuv, dtype="<u2" if bd > 8 else np.uint8)
    assert rec.size == per * frames, (rec.size, per, frames, r.stdout[-2000:])
    return rec, r.stdout


def save(name, stream, rec, w, h, frames, bd):
    out = {"bitstream": np.frombuffer(stream, dtype=np.uint8), "geom": np.array([w, h, frames, bd], dtype=np.int32)}
    per = w * h * 3 // 2
    for poc in range(frames):
        fr = rec[poc * per:(poc + 1) * per].astype(np.int16)
        out["poc%02d_0" % poc] = fr[:w * h].reshape(h, w)
        out["poc%02d_1" % poc] = fr[w * h:w * h * 5 // 4].reshape(h // 2, w // 2)
        out["poc%02d_2" % poc] = fr[w * h * 5 // 4:].reshape(h // 2, w // 2)
    path = os.path.join(mg.GOLD, "lite_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB, %d bytes of bitstream)" % (path, os.path.getsize(path) / 1024.0, len(stream)))


def join(nals):
    return b"".join(b"\x00\x00\x00\x01" + n for n in nals)


def nal_type(n):
    return (n[0] >> 1) & 0x3f


def surgery(name, base, transform):
    with tempfile.TemporaryDirectory() as tmp:
        bs, enc_rec, (w, h, frames, bd) = mg.encode(base, tmp)
        nals = [n for n in split_nals(bs) if nal_type(n) not in (39, 40)]     # the hash SEIs describe the encoder's pictures: dropped
        new = transform(nals)
        stream = join(new)
        rec, log = decode_with_hm(stream, w, h, frames, bd, tmp, name)
        plain, _ = decode_with_hm(join(nals), w, h, frames, bd, tmp, name + "_plain")
        differ = int((rec != plain).sum())
        print("%s: %d of %d samples differ from the untouched stream's pictures" % (name, differ, rec.size))
        assert differ > 0, "the surgery changed nothing"
        save(name, stream, rec, w, h, frames, bd)


def pps_scaling_lists(nals):
    rng = random.Random(20262)
    out = []
    for n in nals:
        if nal_type(n) == 34:
            bits = strip_trailing(nal_bits(n))
            p = parse_pps(bits)
            pos = p["pos_scaling_list_present"]
            bits = bits[:pos] + "1" + random_scaling_list_data(rng) + bits[pos + 1:]
            n = make_nal(n[:2], with_trailing(bits))
        elif nal_type(n) == 33:
            assert parse_sps(strip_trailing(nal_bits(n)))["scaling_list_enabled"], "the base stream must enable scaling lists (ScalingList=1)"
        out.append(n)
    return out


def long_term_and_modification(nals):
    out, sps, pps = [], None, None
    state = {"prev_lsb": 0, "prev_msb": 0, "log": []}
    for n in nals:
        t = nal_type(n)
        if t == 33:
            bits = strip_trailing(nal_bits(n))
            sps = parse_sps(bits)
            pos = sps["pos_lt_present"]
            bits = bits[:pos] + "1" + w_ue(0) + bits[pos + 1:]             # long_term_ref_pics_present_flag, num_long_term_ref_pics_sps = 0
            n = make_nal(n[:2], with_trailing(bits))
        elif t == 34:
            bits = strip_trailing(nal_bits(n))
            pps = parse_pps(bits)
            pos = pps["pos_lists_modification"]
            bits = bits[:pos] + "1" + bits[pos + 1:]
            n = make_nal(n[:2], with_trailing(bits))
        elif t < 32:
            n = make_nal(n[:2], rewrite_slice_header(nal_bits(n), t, sps, pps, state, lt_poc=1))
        out.append(n)
    for e in state["log"]:
        print("  POC %2d  short-term %s  long-term %s  modification %s" % e)
    return out


def main():
    mg.STREAMS.update(mg.LITE)
    mg.STREAMS["surgery_base_sl"] = ("encoder_lowdelay_P_main.cfg", 208, 120, 6, 8, 8, 30, ["--ScalingList=1"])
    mg.STREAMS["surgery_base_ldp"] = ("encoder_lowdelay_P_main10.cfg", 208, 120, 10, 10, 10, 30, [])
    surgery("surgery_ppssl_main8_208x120", "surgery_base_sl", pps_scaling_lists)
    surgery("surgery_ltr_rplm_main10_208x120", "surgery_base_ldp", long_term_and_modification)


if __name__ == "__main__":
    main()
