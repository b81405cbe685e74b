// micro-benchmark: does a 16-byte-per-lane load cost the issuing SIMD VALU time when it RETURNS into VGPRs?  The same scattered-row
// loads + packed-dot VALU work as overlaptest.hip, with the loads (a) into VGPRs, (b) into LDS by global_load_lds (no VGPR write-back),
// (c) into LDS by global_load_lds and then read back with ds_read_b128.  8 waves/SIMD, data L1/L2-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef short short2v __attribute__((ext_vector_type(2)));
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
// MODE bit 0: VALU work; bits 1-2: 0 no loads, 1 loads to VGPRs, 2 loads to LDS, 3 loads to LDS + ds_read
template <int MODE>
__global__ void __launch_bounds__(256) k(const char* buf, uint32_t* out, int iters, int nalu) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4][4][256];     // [wave][load][lane * 4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = buf + (size_t)(blockIdx.x % 64) * 7936 * 64;
  constexpr int G = 8, LD = (MODE >> 1) & 3;
  const int grp = lane / G, in = lane % G;
  uint32_t acc[8];
  for (int i = 0; i < 8; i++) acc[i] = lane * 77 + i;
  uint32_t sink = 0;
  for (int it = 0; it < iters; it++) {
    u32x4 v[4] = {};
    if (LD) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int row = (grp + (it * 4 + u + wave * 3) * (64 / G)) % 64;
        const char* p = base + (size_t)row * 7936 + in * 16;
        if (LD == 1) v[u] = *(const u32x4 AS1*)p;
        else __builtin_amdgcn_global_load_lds((const void AS1*)p, (void AS3*)&lds[wave][u][0], 16, 0, 0);
      }
    }
    if (MODE & 1) {
      for (int a = 0; a < nalu; a++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, acc[(i + 1) & 7]), __builtin_bit_cast(short2v, 0x00030005u), (int)acc[i], false);
    }
    if (LD >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (LD == 3) {
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const u32x4*>(&lds[wave][u][lane * 4]);
    }
    if (LD == 1 || LD == 3) {
#pragma unroll
      for (int u = 0; u < 4; u++) sink += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (LD == 2) sink += lds[wave][0][(lane * 4 + it) & 255];
  }
  uint32_t r = sink;
  for (int i = 0; i < 8; i++) r ^= acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> float run(const char* buf, uint32_t* out, int nalu) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 300, blocks = 256 * 8;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, nalu);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, nalu);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  char* buf; uint32_t* out;
  hipMalloc(&buf, 64 * 7936 * 64 + 4096); hipMalloc(&out, 256 * 8 * 256 * 4); hipMemset(buf, 1, 64 * 7936 * 64 + 4096);
  for (int nalu : {4, 8, 16}) {
    printf("per iteration: 4 loads (16 B/lane, groups of 8 lanes per row) + %d x 8 dot2 per wave\n", nalu);
    printf("  alu alone %.3f ms\n", run<1>(buf, out, nalu));
    printf("  loads to VGPRs        : alone %.3f  with alu %.3f ms\n", run<2>(buf, out, nalu), run<3>(buf, out, nalu));
    printf("  loads to LDS (DMA)    : alone %.3f  with alu %.3f ms\n", run<4>(buf, out, nalu), run<5>(buf, out, nalu));
    printf("  DMA + ds_read_b128    : alone %.3f  with alu %.3f ms\n", run<6>(buf, out, nalu), run<7>(buf, out, nalu));
  }
  return 0;
}
