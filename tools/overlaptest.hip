// micro-benchmark: do scattered 16-byte-per-lane loads (TA-bound, L1/L2-resident data) overlap with packed-dot VALU work of other waves?
// mode 1: VALU only, mode 2: loads only, mode 3: both in the same wave (loads issued, ALU work, then the loaded data consumed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef short short2v __attribute__((ext_vector_type(2)));
template <int MODE, int G>
__global__ void __launch_bounds__(256) k(const char* buf, uint32_t* out, int iters, int nalu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = buf + (size_t)(blockIdx.x % 64) * 7936 * 64 + 4;
  const int grp = lane / G, in = lane % G;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = lane * 77 + i;
  uint32_t sink = 0;
  for (int it = 0; it < iters; it++) {
    u32x4 v[4];
    if (MODE & 2) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int row = (grp + (it * 4 + u + wave * 3) * (64 / G)) % 64;
        v[u] = *(const u32x4_a4*)(base + (size_t)row * 7936 + in * 16);
      }
    }
    if (MODE & 1) {
      for (int a = 0; a < nalu; a++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, acc[(i + 1) & 7]), __builtin_bit_cast(short2v, 0x00030005u), (int)acc[i], false);
    }
    if (MODE & 2) {
#pragma unroll
      for (int u = 0; u < 4; u++) sink += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  uint32_t r = sink;
  for (int i = 0; i < 8; i++) r ^= acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE, int G> float run(const char* buf, uint32_t* out, int nalu) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 300, blocks = 256 * 8;
  hipLaunchKernelGGL((k<MODE, G>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, nalu);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, G>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, nalu);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  char* buf; uint32_t* out;
  hipMalloc(&buf, 64 * 7936 * 64 + 4096); hipMalloc(&out, 256 * 8 * 256 * 4); hipMemset(buf, 1, 64 * 7936 * 64 + 4096);
  for (int nalu : {4, 8, 16}) {
    printf("per iteration: 4 loads (16 B/lane) + %d x 8 dot2 per wave, 8 waves/SIMD\n", nalu);
    printf("  groups of 8 lanes: alu %.3f  loads %.3f  both %.3f ms\n", run<1, 8>(buf, out, nalu), run<2, 8>(buf, out, nalu), run<3, 8>(buf, out, nalu));
    printf("  groups of 2 lanes: alu %.3f  loads %.3f  both %.3f ms\n", run<1, 2>(buf, out, nalu), run<2, 2>(buf, out, nalu), run<3, 2>(buf, out, nalu));
    printf("  single lanes     : alu %.3f  loads %.3f  both %.3f ms\n", run<1, 1>(buf, out, nalu), run<2, 1>(buf, out, nalu), run<3, 1>(buf, out, nalu));
  }
  return 0;
}
