// micro-benchmark: L1/TA throughput of per-lane 16-byte loads for the lane patterns the MC kernel uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4_al __attribute__((ext_vector_type(4)));
typedef u32x4_al u32x4 __attribute__((aligned(2)));
#define AS1 __attribute__((address_space(1)))
// mode 0: 8 lanes across (16 B apart) x 8 rows, offset `off` samples; mode 1: 64 lanes across contiguous
__global__ void k(const int16_t* p, int pitch, int off, int rows, int mode, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int16_t* base;
  if (mode == 0) base = p + (size_t)((wave % 32) * 64 + (lane >> 3) * 8) * pitch + (wave / 32 % 30) * 128 + (lane & 7) * 8 + off;
  else base = p + (size_t)((wave % 32) * 64) * pitch + (wave / 32 % 4) * 512 + lane * 8 + off;
  uint32_t acc = 0;
  for (int r = 0; r < rows; r++) {
    const u32x4 v = *(const u32x4 AS1*)(base + (size_t)(r % 15) * pitch);
    const u32x4 w = *(const u32x4 AS1*)(base + (size_t)(r % 15) * pitch + 8);
    acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w;
  }
  if (acc == 0x12345678) out[0] = acc;
}
int main() {
  const int pitch = 4160, H = 2320;
  int16_t* d; uint32_t* o;
  hipMalloc(&d, (size_t)pitch * H * 2); hipMalloc(&o, 4);
  hipMemset(d, 1, (size_t)pitch * H * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int rows = 15 * 40, blocks = 2048;
  for (int mode = 0; mode < 2; mode++)
    for (int off : {0, 2, 4, 1, 3, 7}) {

      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, pitch, off, rows, mode, o);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, pitch, off, rows, mode, o);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes = (double)blocks * 256 * rows * 32;
      printf("mode %d off %d: %.3f ms  %.1f GB/s lane-bytes  (%.1f B/clk/CU at 2.4GHz)\n", mode, off, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
    }
  return 0;
}
