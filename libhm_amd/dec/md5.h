// md5.h -- RFC 1321 message digest, for the MD5 variant of the decoded-picture-hash SEI (Rec. ITU-T H.265 D.3.19; HM: libmd5/).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace hmdec {

class Md5 {
 public:
  Md5() { a_ = 0x67452301u; b_ = 0xefcdab89u; c_ = 0x98badcfeu; d_ = 0x10325476u; }
  void update(const uint8_t* p, size_t n) {
    total_ += n;
    while (n) {
      if (fill_ == 0 && n >= 64) { block(p); p += 64; n -= 64; continue; }     // whole blocks straight from the caller's buffer
      const size_t take = n < 64 - fill_ ? n : 64 - fill_;
      memcpy(buf_ + fill_, p, take);
      fill_ += take; p += take; n -= take;
      if (fill_ == 64) { block(buf_); fill_ = 0; }
    }
  }
  void final(uint8_t out[16]) {
    const uint64_t bits = total_ * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill_ != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (8 * i));
    update(len, 8);
    const uint32_t s[4] = {a_, b_, c_, d_};
    for (int i = 0; i < 16; i++) out[i] = (uint8_t)(s[i >> 2] >> (8 * (i & 3)));
  }

 private:
  static uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const uint8_t S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    memcpy(m, p, 64);                    // (little-endian host, as everything else in this decoder)
    uint32_t a = a_, b = b_, c = c_, d = d_;
    // four rounds of sixteen steps; the message index and the rotation of a step are compile-time constants once unrolled
#define HMDEC_MD5_STEP(F, i, g)                                    \
    { const uint32_t f = (F), t = d; d = c; c = b; b = b + rol(a + f + K[i] + m[g], S[i]); a = t; }
#pragma GCC unroll 16
    for (int i = 0; i < 16; i++) HMDEC_MD5_STEP((b & c) | (~b & d), i, i)
#pragma GCC unroll 16
    for (int i = 16; i < 32; i++) HMDEC_MD5_STEP((d & b) | (~d & c), i, (5 * i + 1) & 15)
#pragma GCC unroll 16
    for (int i = 32; i < 48; i++) HMDEC_MD5_STEP(b ^ c ^ d, i, (3 * i + 5) & 15)
#pragma GCC unroll 16
    for (int i = 48; i < 64; i++) HMDEC_MD5_STEP(c ^ (b | ~d), i, (7 * i) & 15)
#undef HMDEC_MD5_STEP
    a_ += a; b_ += b; c_ += c; d_ += d;
  }
  uint32_t a_, b_, c_, d_;
  uint64_t total_ = 0;
  uint8_t buf_[64];
  size_t fill_ = 0;
};

}  // namespace hmdec
