// micro-benchmark: cost of one wave-wide 16-byte-per-lane load as a function of how the 64 lane addresses spread over cache lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4_al __attribute__((ext_vector_type(4)));
typedef u32x4_al u32x4 __attribute__((aligned(4)));
#define AS1 __attribute__((address_space(1)))
// group = lanes that read adjacent 16-byte pieces (a PU's width / 8); every group sits at a pseudo-random place
__global__ void k(const int16_t* p, int pitch, int group, int rows, uint32_t seed, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int g = lane / group, i = lane % group;
  uint32_t h = (wave * 64 + g) * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const int x = (h % 3600) & ~1, y = (h >> 12) % 2000;
  const int16_t* base = p + (size_t)y * pitch + x + i * 8;
  uint32_t acc = 0;
  for (int r = 0; r < rows; r++) {
    const u32x4 v = *(const u32x4 AS1*)(base + (size_t)(r % 15) * pitch);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678) out[0] = acc;
}
int main() {
  const int pitch = 4160, H = 2320;
  int16_t* d; uint32_t* o;
  hipMalloc(&d, (size_t)pitch * H * 2); hipMalloc(&o, 4);
  hipMemset(d, 1, (size_t)pitch * H * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int rows = 15 * 20, blocks = 2048;
  for (int group : {64, 16, 8, 4, 2, 1}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, pitch, group, rows, 1u, o);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, pitch, group, rows, 2u, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double loads = (double)blocks * 4 * rows;
    printf("group %2d: %.3f ms  %.1f clk per wave-load per CU (2.4 GHz, 256 CUs)\n", group, ms, ms * 1e-3 * 2.4e9 * 256 / loads);
  }
  return 0;
}
