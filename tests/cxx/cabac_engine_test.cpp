// The CABAC engine of libhmdec (libhm_amd/dec/cabac.h): grouped bypass decoding (reciprocal multiplication, 16 bins read ahead and
// partly given back) against the bin-by-bin formulation of 9.3.4.3.4, on random data.  Built and run by tests/test_cabac_engine.py.
#include "cabac.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace hmdec;
int main() {
  std::vector<uint8_t> d(4096);
  srand(1);
  for (auto& b : d) b = rand() & 0x7f;
  // reference: bit-by-bit bypass vs batch
  for (int trial = 0; trial < 2000; trial++) {
    Cabac a, b;
    a.attach(d.data(), d.size()); b.attach(d.data(), d.size());
    a.start(trial * 8 % 8000); b.start(trial * 8 % 8000);
    ctx_t c1 = (ctx_t)(rand() % 126), c2 = c1;
    for (int k = 0; k < rand() % 5; k++) { int x = a.decision(c1), y = b.decision(c2); if (x != y) { printf("decision mismatch\n"); return 1; } }
    for (int rep = 0; rep < 20; rep++) {
      int m = rand() % 17;
      unsigned ref = 0;
      for (int i = 0; i < m; i++) ref = (ref << 1) | (unsigned)a.bypass();
      unsigned wide; unsigned q = b.bypass_peek16(wide);
      b.bypass_keep(wide, q, m);
      unsigned got = m ? q >> (16 - m) : 0;
      if (got != ref || a.bit_pos() != b.bit_pos()) { printf("trial %d rep %d m %d: ref %x got %x pos %zu %zu\n", trial, rep, m, ref, got, a.bit_pos(), b.bit_pos()); return 1; }
      int x = a.decision(c1), y = b.decision(c2);
      if (x != y) { printf("decision after bypass mismatch trial %d rep %d\n", trial, rep); return 1; }
      int n = 1 + rand() % 16;
      unsigned r2 = 0; for (int i = 0; i < n; i++) r2 = (r2 << 1) | (unsigned)a.bypass();
      unsigned g2 = b.bypass_bits(n);
      if (r2 != g2) { printf("bypass_bits mismatch n %d %x %x\n", n, r2, g2); return 1; }
    }
  }
  printf("ok\n");
  return 0;
}
