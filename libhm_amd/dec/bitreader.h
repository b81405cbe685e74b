// bitreader.h -- NAL unit payload access for the host-side HEVC parser.
// Rec. ITU-T H.265 7.3.1.1 (emulation prevention), 7.2 (read_bits), 9.2 (Exp-Golomb).  HM counterpart:
// TLibDecoder/NALread.cpp:50-110 (convertPayloadToRBSP), TComBitStream.cpp, TDecCAVLC.cpp xReadUvlc/xReadSvlc.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace hmdec {

struct ParseError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// NAL payload (after the two header bytes) -> RBSP: 0x000003 -> 0x0000
// `removed` (optional): payload positions of the emulation prevention bytes that were dropped -- entry point offsets in the slice
// header count them (7.4.7.1), positions in the RBSP do not
inline std::vector<uint8_t> nal_to_rbsp(const uint8_t* p, size_t n, std::vector<size_t>* removed = nullptr) {
  std::vector<uint8_t> out;
  out.reserve(n);
  int zeros = 0;
  for (size_t i = 0; i < n; i++) {
    if (zeros >= 2 && p[i] == 3) { zeros = 0; if (removed) removed->push_back(i); continue; }
    out.push_back(p[i]);
    zeros = p[i] == 0 ? zeros + 1 : 0;
  }
  return out;
}

class BitReader {
 public:
  BitReader(const uint8_t* p, size_t bytes) : p_(p), bits_(bytes * 8) {}
  size_t pos() const { return pos_; }
  size_t size_bits() const { return bits_; }
  bool byte_aligned() const { return (pos_ & 7) == 0; }
  uint32_t u(int n) {
    if (n == 0) return 0;
    if (pos_ + n > bits_) throw ParseError("read past the end of the RBSP");
    uint32_t v = 0;
    while (n > 0) {
      const int avail = 8 - (int)(pos_ & 7), take = n < avail ? n : avail;
      v = (v << take) | ((p_[pos_ >> 3] >> (avail - take)) & ((1u << take) - 1));
      pos_ += take;
      n -= take;
    }
    return v;
  }
  bool flag() { return u(1) != 0; }
  uint32_t ue() {
    int zeros = 0;
    while (!u(1)) if (++zeros > 32) throw ParseError("Exp-Golomb prefix too long");
    return zeros == 0 ? 0 : (uint32_t)(((uint64_t)1 << zeros) - 1 + (zeros == 32 ? 0 : u(zeros)));
  }
  int32_t se() {
    const uint32_t k = ue();
    return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
  }
  void skip(size_t n) {
    if (pos_ + n > bits_) throw ParseError("skip past the end of the RBSP");
    pos_ += n;
  }
  // 7.2 more_rbsp_data(): anything before the last 1 bit of the RBSP
  bool more_rbsp_data() const {
    if (pos_ >= bits_) return false;
    size_t last = bits_;
    while (last > pos_) {
      const size_t b = last - 1;
      if ((p_[b >> 3] >> (7 - (b & 7))) & 1) break;
      last--;
    }
    return last >= pos_ + 2;           // the last 1 is rbsp_stop_one_bit: data only if it lies beyond the current bit
  }
  void byte_alignment() {               // 7.3.2.5: a one and zeros up to the next byte
    if (!u(1)) throw ParseError("alignment bit is not 1");
    while (!byte_aligned()) if (u(1)) throw ParseError("alignment bits are not 0");
  }
  const uint8_t* data() const { return p_; }

 private:
  const uint8_t* p_;
  size_t bits_, pos_ = 0;
};

}  // namespace hmdec
