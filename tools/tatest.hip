// micro-benchmark: throughput of the vector-memory path (TA/TCP) for 16-byte-per-lane loads by access shape, data resident in L1/L2.
//   shape: lanes in groups of G consecutive 16-byte pieces, one group per row (row stride 7936 B), start of a group misaligned by `mis` bytes
// prints cycles per wave-instruction per CU (all CUs busy, 8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
template <int BYTES>
__global__ void __launch_bounds__(256) k(const char* buf, uint32_t* out, int G, int mis, int rows_span, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each wave walks `rows_span` rows cyclically so that the footprint per CU stays L1-sized
  const char* base = buf + (size_t)(blockIdx.x % 64) * 7936 * 64 + mis;
  const int grp = lane / G, in = lane % G;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int row = (grp + (it * 8 + u + wave * 3) * (64 / G)) % rows_span;
      const char* p = base + (size_t)row * 7936 + in * BYTES;
      if (BYTES == 16) { const u32x4 v = *(const u32x4_a4*)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
      else if (BYTES == 8) { const u32x2 v = *(const u32x2_a4*)p; acc += v.x ^ v.y; }
      else { acc += *(const uint32_t*)p; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int BYTES> void run(const char* buf, uint32_t* out, int G, int mis, int rows_span) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 400, blocks = 256 * 8;
  hipLaunchKernelGGL(k<BYTES>, dim3(blocks), dim3(256), 0, 0, buf, out, G, mis, rows_span, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<BYTES>, dim3(blocks), dim3(256), 0, 0, buf, out, G, mis, rows_span, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr_per_cu = (double)blocks * 4 * iters * 8 / 256;
  printf("%2d B/lane  group %2d  misalign %2d  rows %3d : %6.1f cycles per wave-instr per CU (2.4 GHz)  = %5.1f B/clk/CU\n", BYTES, G, mis, rows_span,
         ms * 1e-3 * 2.4e9 / winstr_per_cu, 64.0 * BYTES / (ms * 1e-3 * 2.4e9 / winstr_per_cu));
}
int main() {
  char* buf; uint32_t* out;
  hipMalloc(&buf, 64 * 7936 * 64 + 4096); hipMalloc(&out, 256 * 8 * 256 * 4); hipMemset(buf, 1, 64 * 7936 * 64 + 4096);
  for (int rows : {16, 64}) {
    for (int G : {64, 8, 1}) {
      if (G == 64 && rows != 16) continue;
      for (int mis : {0, 4, 8, 2 * 4 + 16}) run<16>(buf, out, G, mis, rows);
    }
  }
  for (int G : {2, 3, 4}) for (int mis : {0, 4, 40, 104}) run<16>(buf, out, G, mis, 64);   // pairs / triples / quads of lanes per row (104: the group crosses a 128-byte line)
  for (int mis : {0, 4}) { run<8>(buf, out, 8, mis, 16); run<4>(buf, out, 8, mis, 16); run<8>(buf, out, 16, mis, 16); run<4>(buf, out, 32, mis, 16); }
  return 0;
}
