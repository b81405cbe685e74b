// cabac.h -- CABAC parsing process of Rec. ITU-T H.265 9.3: arithmetic decoding engine (9.3.4.3), context variables and their
// initialisation (9.3.2.2, Tables 9-5 ... 9-37).  HM counterpart: TDecBinCoderCABAC.cpp, ContextModel.cpp, ContextTables.h.
// The engine is the specification's own formulation: a 9-bit offset register refilled bit by bit from a position counter, so the
// places where the syntax goes back to plain bits (pcm_sample, byte_alignment after end_of_subset_one_bit) need no rewinding.
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
  CTX_CBF_CHROMA = CTX_CBF_LUMA + 2,        // 5 (by transform depth; depth 4: 4x4 chroma blocks of 4:4:4)
  CTX_MVD_GT0 = CTX_CBF_CHROMA + 5,         // 1
  CTX_MVD_GT1 = CTX_MVD_GT0 + 1,            // 1
  CTX_QP_DELTA = CTX_MVD_GT1 + 1,           // 2
  CTX_TS_FLAG = CTX_QP_DELTA + 2,           // 2 (luma, chroma)
  CTX_LAST_X = CTX_TS_FLAG + 2,             // 18
  CTX_LAST_Y = CTX_LAST_X + 18,             // 18
  CTX_CSBF = CTX_LAST_Y + 18,               // 4
  CTX_SIG = CTX_CSBF + 4,                   // 42 + the two transform_skip_context_enabled contexts (luma 42, chroma 43)
  CTX_GT1 = CTX_SIG + 44,                   // 24
  CTX_GT2 = CTX_GT1 + 24,                   // 6
  CTX_RDPCM_FLAG = CTX_GT2 + 6,             // 2 (luma, chroma): explicit_rdpcm_flag
  CTX_RDPCM_DIR = CTX_RDPCM_FLAG + 2,       // 2: explicit_rdpcm_dir_flag
  CTX_CCP = CTX_RDPCM_DIR + 2,              // 10: per chroma component log2_res_scale_abs_plus1 (4, by bin) + res_scale_sign_flag (HM's packing, ContextTables.h:493)
  CTX_COUNT = CTX_CCP + 10
};

// (a context variable is 7 bits; it is kept in 16 so that stores to it cannot alias the engine's registers -- unsigned char may
// alias anything, which would force the compiler to reload range and offset after every bin)
typedef uint16_t ctx_t;
struct ContextSet {
  ctx_t s[CTX_COUNT];                       // (pStateIdx << 1) | valMps
  uint8_t stat_coeff[4] = {0, 0, 0, 0};     // persistent_rice_adaptation: StatCoeff[2 * chroma + (skip | bypass)], synchronised and
                                            // reset with the context variables (9.3.2.4; HM: TDecSbac.cpp:169, 1858)
  void init(int init_type, int slice_qp);   // 9.3.2.2
};

class Cabac {
 public:
  void attach(const uint8_t* rbsp, size_t bytes) { p_ = rbsp; nbytes_ = bytes; }
  size_t bit_pos() const { return next_byte_ * 8 - avail_; }
  // 9.3.2.5: initialisation of the arithmetic decoding engine at a byte-aligned position
  void start(size_t bit_pos) {
    next_byte_ = bit_pos >> 3;
    avail_ = 0;
    res_ = 0;
    if (bit_pos & 7) get((int)(bit_pos & 7));
    range_ = 510;
    offset_ = get(9);
  }
  // 9.3.4.3.2, without a data-dependent branch: on coded video the bin is what the branch predictor cannot know, and a
  // mispredicted branch costs as much as the rest of the bin.  MPS and LPS path are computed together and selected by a mask, the state
  // transition is one table indexed by (state, MPS, LPS taken), renormalisation shifts by the leading-zero count (0 when none is due).
  int decision(ctx_t& ctx) {
    const unsigned s = ctx;                                  // (pStateIdx << 1) | valMps
    const unsigned lps = kRangeLps[s >> 1][(range_ >> 6) & 3];
    const unsigned rmps = range_ - lps;
    const unsigned is_lps = offset_ >= rmps ? 1u : 0u;
    const unsigned mask = 0u - is_lps;
    offset_ -= rmps & mask;
    const unsigned r = rmps ^ ((rmps ^ lps) & mask);
    ctx = kNext[(s << 1) | is_lps];
    const int n = __builtin_clz(r) - 23;                    // shifts until bit 8 is set (0 for an MPS range >= 256)
    range_ = r << n;
    offset_ = (offset_ << n) | get(n);
    return (int)((s & 1u) ^ is_lps);
  }
  int bypass() {
    offset_ = (offset_ << 1) | get(1);
    const unsigned b = offset_ >= range_ ? 1u : 0u;
    offset_ -= range_ & (0u - b);
    return (int)b;
  }
  // n bypass bins at once (n <= 16): with offset < range before, the n bins are the quotient of (offset << n | new bits) by range.
  // The range only changes in decision(), so the division is a multiplication by a tabulated reciprocal (exact: the numerator has
  // 25 bits at most, the reciprocal 39 fractional bits).
  unsigned bypass_bits(int n) {
    if (n <= 0) return 0;
    unsigned v = 0;
    while (n > 0) {
      const int k = n > 16 ? 16 : n;
      const unsigned wide = (offset_ << k) | get(k);          // offset < range <= 510: 25 bits at most
      const unsigned q = (unsigned)(((uint64_t)wide * kInv[range_]) >> 39);
      offset_ = wide - q * range_;
      v = (v << k) | q;
      n -= k;
    }
    return v;
  }
  // The next 16 bypass bins (bit 15 = the first) WITHOUT committing to them: bypass bins are a bit sequence whatever way they are
  // grouped, so a syntax element of unknown length (coeff_abs_level_remaining) is read as one group and the engine then keeps the
  // first m bins only -- floor(floor(W / 2^k) / r) = floor(floor(W / r) / 2^k), i.e. the state after m bins follows from the group's
  // numerator and quotient, and the bit position steps back by the rest.
  unsigned bypass_peek16(unsigned& wide) {
    wide = (offset_ << 16) | get(16);
    return (unsigned)(((uint64_t)wide * kInv[range_]) >> 39);
  }
  void bypass_keep(unsigned wide, unsigned q, int m) {         // after bypass_peek16: m <= 16 of the bins are consumed
    const int back = 16 - m;
    offset_ = (wide >> back) - (q >> back) * range_;
    avail_ += back;
  }
  // 9.3.4.3.5; when the result is 1 the engine is finished: call finish_to_byte() before reading plain bits or restarting
  int terminate() {
    range_ -= 2;
    if (offset_ >= range_) return 1;
    if (range_ < 256) { range_ <<= 1; offset_ = (offset_ << 1) | get(1); }
    return 0;
  }
  // After a terminating bin equal to 1 the engine has read exactly up to and including the 1 bit that rbsp_trailing_bits() /
  // byte_alignment() start with (the encoder's flush, 9.3.4.5, writes it as its last bit; the first bit the encoder would have
  // put out is suppressed, which is what makes the 9-bit window end there).  What follows are zero bits up to the byte boundary;
  // pcm_sample data or the next sub-stream start at that boundary.
  void finish_to_byte() {
    const size_t pos = bit_pos();
    if (pos == 0 || pos > nbytes_ * 8 || !((p_[(pos - 1) >> 3] >> (7 - ((pos - 1) & 7))) & 1)) throw ParseError("CABAC: no stop bit behind a terminating bin");
    while (bit_pos() & 7) if (get(1)) throw ParseError("CABAC: alignment bits are not zero");
  }
  unsigned plain_bits(int n) { return get(n); }     // pcm_sample_*: read_bits(n) between finish_to_byte() and start()

 private:
  unsigned get(int n) {                               // n <= 25 (0 allowed); bits behind the end of the data read as zero
    if (avail_ < 32) refill();                        // (taken once in 32 bits: the one branch of a bin, and a predictable one)
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
      while (avail_ <= 48) {                           // (at most 56 bits in hand: a shift by avail_ is always defined, get(0) included)
        res_ = (res_ << 8) | (next_byte_ < nbytes_ ? p_[next_byte_] : 0u);
        next_byte_++;
        avail_ += 8;
      }
      if (next_byte_ > nbytes_ + 24) throw ParseError("CABAC: read far past the end of the slice data");
    }
  }
  const uint8_t* p_ = nullptr;
  size_t nbytes_ = 0, next_byte_ = 0;
  uint64_t res_ = 0;
  int avail_ = 0;
  unsigned range_ = 510, offset_ = 0;
  static const uint8_t kRangeLps[64][4];
  static const uint8_t kNextLps[64], kNextMps[64];
  static const ctx_t kNext[256];
  static const uint32_t kInv[512];                    // [range]: floor(2^39 / range) + 1 for range >= 256 (bypass_bits)                      // [(state << 1 | mps) << 1 | LPS taken] -> next (state << 1 | mps)
};

}  // namespace hmdec
