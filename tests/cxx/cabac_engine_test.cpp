// The CABAC engine of libhmdec (libhm_amd/dec/cabac.h) against a literal transcription of Rec. ITU-T H.265 9.3.4.3 (9-bit offset
// register, read_bits(1) per renormalisation step, bin-by-bin bypass), on random data with random context states: context-coded
// bins, single bypass bins, grouped bypass bins (reciprocal multiplication), 16 bins read ahead and partly given back, terminate
// bins, and the bit position after each.  Built and run by tests/test_cabac_engine.py.
#include "cabac.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace hmdec;

static const uint8_t kLpsRef[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185},
    {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},  {85, 104, 123, 142},  {81, 99, 117, 135},
    {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},   {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},
    {56, 69, 81, 94},     {53, 65, 77, 89},     {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},
    {41, 50, 59, 69},     {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},     {23, 28, 33, 39},
    {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},     {18, 22, 26, 30},     {17, 21, 25, 28},
    {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},     {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},
    {12, 14, 17, 20},     {11, 14, 16, 19},     {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},
    {9, 11, 12, 14},      {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
static const uint8_t kNextLpsRef[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
                                        18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
                                        31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};
static const uint8_t kNextMpsRef[64] = {1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
                                        23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44,
                                        45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 62, 63};

struct Ref {                          // 9.3.4.3 as written
  const uint8_t* p; size_t n, pos;
  unsigned range, offset;
  unsigned bit() { const unsigned b = pos < n * 8 ? (p[pos >> 3] >> (7 - (pos & 7))) & 1 : 0; pos++; return b; }
  void start(size_t at) { pos = at; range = 510; offset = 0; for (int i = 0; i < 9; i++) offset = (offset << 1) | bit(); }
  int decision(unsigned& state, unsigned& mps) {
    const unsigned lps = kLpsRef[state][(range >> 6) & 3];
    range -= lps;
    int bin;
    if (offset >= range) { bin = !mps; offset -= range; range = lps; if (state == 0) mps = 1 - mps; state = kNextLpsRef[state]; }
    else { bin = mps; state = kNextMpsRef[state]; }
    while (range < 256) { range <<= 1; offset = (offset << 1) | bit(); }
    return bin;
  }
  int bypass() { offset = (offset << 1) | bit(); if (offset >= range) { offset -= range; return 1; } return 0; }
  int terminate() { range -= 2; if (offset >= range) return 1; while (range < 256) { range <<= 1; offset = (offset << 1) | bit(); } return 0; }
};

int main() {
  std::vector<uint8_t> d(1 << 16);
  srand(1);
  for (auto& b : d) b = (uint8_t)rand();
  long checked = 0;
  for (int trial = 0; trial < 4000; trial++) {
    const size_t at = (size_t)(rand() % 60000) * 8;
    if ((((unsigned)d[at >> 3] << 1) | (d[(at >> 3) + 1] >> 7)) >= 510) continue;     // not a valid initial offset
    Ref a{d.data(), d.size(), 0, 0, 0};
    Cabac b;
    a.start(at);
    b.attach(d.data(), d.size());
    b.start(at);
    unsigned st[8], mp[8];
    ctx_t cx[8];
    for (int i = 0; i < 8; i++) { st[i] = rand() % 63; mp[i] = rand() & 1; cx[i] = (ctx_t)((st[i] << 1) | mp[i]); }
    for (int step = 0; step < 400; step++) {
      const int kind = rand() % 10;
      if (kind < 5) {
        const int i = rand() % 8;
        const int x = a.decision(st[i], mp[i]), y = b.decision(cx[i]);
        if (x != y || cx[i] != (ctx_t)((st[i] << 1) | mp[i])) { printf("decision: trial %d step %d\n", trial, step); return 1; }
      } else if (kind == 5) {
        if (a.bypass() != b.bypass()) { printf("bypass: trial %d step %d\n", trial, step); return 1; }
      } else if (kind == 6) {
        const int n = 1 + rand() % 20;
        unsigned r = 0;
        for (int i = 0; i < n; i++) r = (r << 1) | (unsigned)a.bypass();
        if (r != b.bypass_bits(n)) { printf("bypass_bits(%d): trial %d step %d\n", n, trial, step); return 1; }
      } else if (kind < 9) {
        const int m = rand() % 17;
        unsigned r = 0;
        for (int i = 0; i < m; i++) r = (r << 1) | (unsigned)a.bypass();
        unsigned wide;
        const unsigned q = b.bypass_peek16(wide);
        b.bypass_keep(wide, q, m);
        if ((m ? q >> (16 - m) : 0u) != r) { printf("peek/keep %d: trial %d step %d\n", m, trial, step); return 1; }
      } else {
        const int x = a.terminate(), y = b.terminate();
        if (x != y) { printf("terminate: trial %d step %d\n", trial, step); return 1; }
        if (x) break;
      }
      if (a.pos != b.bit_pos()) { printf("bit position: trial %d step %d kind %d: %zu vs %zu\n", trial, step, kind, a.pos, b.bit_pos()); return 1; }
      checked++;
    }
  }
  printf("ok\n");
  return checked > 100000 ? 0 : 2;
}
