// cabac.h -- CABAC parsing process of Rec. ITU-T H.265 9.3: arithmetic decoding engine (9.3.4.3), context variables and their
// initialisation (9.3.2.2, Tables 9-5 ... 9-37).  HM counterpart: TDecBinCoderCABAC.cpp, ContextModel.cpp, ContextTables.h.
// The engine follows the specification's formulation bit for bit (same bins, same bit positions), in a form that keeps the stream bits
// behind the offset register in one word with it (class Cabac below).
#pragma once
#include <cstdint>
#include <cstring>

#include "bitreader.h"

namespace hmdec {

// context variable ranges inside the flat table (ctxIdx offsets; counts as in 9.3.4.2)
enum Ctx {
  CTX_SAO_MERGE = 0,                        // 1
  CTX_SAO_TYPE = CTX_SAO_MERGE + 1,         // 1
  CTX_SPLIT_CU = CTX_SAO_TYPE + 1,          // 3
  CTX_TQ_BYPASS = CTX_SPLIT_CU + 3,         // 1
  CTX_SKIP = CTX_TQ_BYPASS + 1,             // 3
  CTX_PRED_MODE = CTX_SKIP + 3,             // 1
  CTX_PART_MODE = CTX_PRED_MODE + 1,        // 4
  CTX_PREV_INTRA = CTX_PART_MODE + 4,       // 1
  CTX_CHROMA_MODE = CTX_PREV_INTRA + 1,     // 1
  CTX_ROOT_CBF = CTX_CHROMA_MODE + 1,       // 1
  CTX_MERGE_FLAG = CTX_ROOT_CBF + 1,        // 1
  CTX_MERGE_IDX = CTX_MERGE_FLAG + 1,       // 1
  CTX_INTER_DIR = CTX_MERGE_IDX + 1,        // 5
  CTX_REF_IDX = CTX_INTER_DIR + 5,          // 2
  CTX_MVP = CTX_REF_IDX + 2,                // 1
  CTX_SPLIT_TU = CTX_MVP + 1,               // 3
  CTX_CBF_LUMA = CTX_SPLIT_TU + 3,          // 2
  CTX_CBF_CHROMA = CTX_CBF_LUMA + 2,        // 4
  CTX_MVD_GT0 = CTX_CBF_CHROMA + 4,         // 1
  CTX_MVD_GT1 = CTX_MVD_GT0 + 1,            // 1
  CTX_QP_DELTA = CTX_MVD_GT1 + 1,           // 2
  CTX_TS_FLAG = CTX_QP_DELTA + 2,           // 2 (luma, chroma)
  CTX_LAST_X = CTX_TS_FLAG + 2,             // 18
  CTX_LAST_Y = CTX_LAST_X + 18,             // 18
  CTX_CSBF = CTX_LAST_Y + 18,               // 4
  CTX_SIG = CTX_CSBF + 4,                   // 42
  CTX_GT1 = CTX_SIG + 42,                   // 24
  CTX_GT2 = CTX_GT1 + 24,                   // 6
  CTX_COUNT = CTX_GT2 + 6
};

// (a context variable is 7 bits; it is kept in 16 so that stores to it cannot alias the engine's registers -- unsigned char may
// alias anything, which would force the compiler to reload range and offset after every bin)
typedef uint16_t ctx_t;
struct ContextSet {
  ctx_t s[CTX_COUNT];                       // (pStateIdx << 1) | valMps
  void init(int init_type, int slice_qp);   // 9.3.2.2
};

// The engine keeps the offset register of 9.3.4.3 together with the bits that follow it in the stream: value_ = ivlOffset * 2^look_ +
// (the next look_ bits).  Comparing ivlOffset with a range is comparing value_ with the range shifted by look_; renormalising by n bits
// -- ivlOffset = ivlOffset << n | read_bits(n) -- leaves value_ as it is and takes n off look_.  So the serial chain of a bin is one
// table look-up, a subtraction, a compare and a leading-zero count for the range, and one masked subtraction for the value; bits enter
// 32 at a time, off that chain.  The bit position the specification's formulation would be at (9 bits for the initial offset plus
// every renormalisation shift) is the fetch position minus look_, which is what pcm_sample, byte_alignment() and the sub-stream
// entry points need.
class Cabac {
 public:
  void attach(const uint8_t* rbsp, size_t bytes) { p_ = rbsp; nbytes_ = bytes; }
  size_t bit_pos() const { return plain_ ? plain_pos_ : next_byte_ * 8 - avail_ - (size_t)look_; }
  // 9.3.2.5: initialisation of the arithmetic decoding engine
  void start(size_t bit_pos) {
    plain_ = false;
    next_byte_ = bit_pos >> 3;
    avail_ = 0;
    res_ = 0;
    if (bit_pos & 7) get((int)(bit_pos & 7));
    range_ = 510;
    value_ = get(9);
    look_ = 0;
  }
  // 9.3.4.3.2, without a data-dependent branch: on coded video the bin is what the branch predictor cannot know, and a
  // mispredicted branch costs as much as the rest of the bin.  MPS and LPS path are computed together and selected by a mask, the state
  // transition is one table indexed by (state, MPS, LPS taken), renormalisation shifts by the leading-zero count (0 when none is due).
  int decision(ctx_t& ctx) {
    if (look_ < 16) more();
    const unsigned s = ctx;                                  // (pStateIdx << 1) | valMps
    const unsigned lps = kRangeLps[s >> 1][(range_ >> 6) & 3];
    const unsigned rmps = range_ - lps;
    const uint64_t scaled = (uint64_t)rmps << look_;
    const unsigned is_lps = value_ >= scaled ? 1u : 0u;
    const unsigned mask = 0u - is_lps;
    value_ -= scaled & (uint64_t)(int64_t)(int32_t)mask;
    const unsigned r = rmps ^ ((rmps ^ lps) & mask);
    ctx = kNext[(s << 1) | is_lps];
    const int n = __builtin_clz(r) - 23;                    // shifts until bit 8 is set (0 for an MPS range >= 256)
    range_ = r << n;
    look_ -= n;
    return (int)((s & 1u) ^ is_lps);
  }
  int bypass() {
    if (look_ < 16) more();
    look_ -= 1;
    const uint64_t scaled = (uint64_t)range_ << look_;
    const unsigned b = value_ >= scaled ? 1u : 0u;
    value_ -= scaled & (uint64_t)(int64_t)(int32_t)(0u - b);
    return (int)b;
  }
  // n bypass bins at once: with offset < range before, the n bins are the quotient of (offset << n | new bits) by range.
  // The range only changes in decision(), so the division is a multiplication by a tabulated reciprocal (exact: the numerator has
  // 25 bits at most, the reciprocal 39 fractional bits).
  unsigned bypass_bits(int n) {
    unsigned v = 0;
    while (n > 0) {
      const int k = n > 16 ? 16 : n;
      if (look_ < 16) more();
      look_ -= k;
      const unsigned wide = (unsigned)(value_ >> look_);      // offset < range <= 510: 25 bits at most
      const unsigned q = (unsigned)(((uint64_t)wide * kInv[range_]) >> 39);
      value_ -= (uint64_t)(q * range_) << look_;
      v = (v << k) | q;
      n -= k;
    }
    return v;
  }
  // The next 16 bypass bins (bit 15 = the first) WITHOUT committing to them: bypass bins are a bit sequence whatever way they are
  // grouped, so a syntax element of unknown length (coeff_abs_level_remaining) is read as one group and the engine then keeps the
  // first m bins only -- floor(floor(W / 2^k) / r) = floor(floor(W / r) / 2^k), i.e. the state after m bins follows from the group's
  // quotient.  Nothing is changed before bypass_keep().
  unsigned bypass_peek16() {
    if (look_ < 16) more();
    return (unsigned)(((uint64_t)(unsigned)(value_ >> (look_ - 16)) * kInv[range_]) >> 39);
  }
  void bypass_keep(unsigned q, int m) {                       // after bypass_peek16: the first m <= 16 of its bins are consumed
    look_ -= m;
    value_ -= (uint64_t)((q >> (16 - m)) * range_) << look_;
  }
  // 9.3.4.3.5; when the result is 1 the engine is finished: call finish_to_byte() before reading plain bits or restarting
  int terminate() {
    if (look_ < 16) more();
    range_ -= 2;
    if (value_ >= ((uint64_t)range_ << look_)) return 1;
    if (range_ < 256) { range_ <<= 1; look_ -= 1; }
    return 0;
  }
  // After a terminating bin equal to 1 the engine has read exactly up to and including the 1 bit that rbsp_trailing_bits() /
  // byte_alignment() start with (the encoder's flush, 9.3.4.5, writes it as its last bit; the first bit the encoder would have
  // put out is suppressed, which is what makes the 9-bit window end there).  What follows are zero bits up to the byte boundary;
  // pcm_sample data or the next sub-stream start at that boundary.
  void finish_to_byte() {
    size_t pos = bit_pos();
    auto bit = [&](size_t i) { return i < nbytes_ * 8 ? (p_[i >> 3] >> (7 - (i & 7))) & 1 : 0; };
    if (pos == 0 || pos > nbytes_ * 8 || !bit(pos - 1)) throw ParseError("CABAC: no stop bit behind a terminating bin");
    while (pos & 7) { if (bit(pos)) throw ParseError("CABAC: alignment bits are not zero"); pos++; }
    plain_ = true;
    plain_pos_ = pos;
  }
  unsigned plain_bits(int n) {                                 // pcm_sample_*: read_bits(n) between finish_to_byte() and start()
    unsigned v = 0;
    for (int i = 0; i < n; i++, plain_pos_++) v = (v << 1) | (plain_pos_ < nbytes_ * 8 ? (unsigned)((p_[plain_pos_ >> 3] >> (7 - (plain_pos_ & 7))) & 1) : 0u);
    return v;
  }

 private:
  void more() {                                               // 32 more bits behind the offset (look_ < 16 before: value_ stays below 2^57)
    value_ = (value_ << 32) | ((uint64_t)get(16) << 16) | get(16);
    look_ += 32;
  }
  unsigned get(int n) {                               // n <= 25 (0 allowed); bits behind the end of the data read as zero
    if (avail_ < 32) refill();
    avail_ -= n;
    return (unsigned)(res_ >> avail_) & ((1u << n) - 1u);
  }
  void refill() {
    if (next_byte_ + 4 <= nbytes_) {
      uint32_t w;
      memcpy(&w, p_ + next_byte_, 4);
      res_ = (res_ << 32) | __builtin_bswap32(w);
      next_byte_ += 4;
      avail_ += 32;
    } else {
      while (avail_ <= 56) {
        res_ = (res_ << 8) | (next_byte_ < nbytes_ ? p_[next_byte_] : 0u);
        next_byte_++;
        avail_ += 8;
      }
      if (next_byte_ > nbytes_ + 24) throw ParseError("CABAC: read far past the end of the slice data");
    }
  }
  const uint8_t* p_ = nullptr;
  size_t nbytes_ = 0, next_byte_ = 0, plain_pos_ = 0;
  uint64_t res_ = 0, value_ = 0;
  int avail_ = 0, look_ = 0;
  unsigned range_ = 510;
  bool plain_ = false;
  static const uint8_t kRangeLps[64][4];
  static const uint8_t kNextLps[64], kNextMps[64];
  static const ctx_t kNext[256];                      // [(state << 1 | mps) << 1 | LPS taken] -> next (state << 1 | mps)
  static const uint32_t kInv[512];                    // [range]: floor(2^39 / range) + 1 for range >= 256 (bypass_bits)
};

}  // namespace hmdec
