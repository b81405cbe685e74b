// micro-benchmark: VALU issue rate of the integer ops the kernels use (wave64, N independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(256) k(const uint32_t* in, uint32_t* out, int iters) {
  uint32_t a[8], acc[8];
  for (int i = 0; i < 8; i++) { a[i] = in[threadIdx.x + 64 * i]; acc[i] = a[i] ^ 0x55; }
  const uint32_t c = in[threadIdx.x + 1024];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) acc[i] = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a[i]), __builtin_bit_cast(short2v, c), (int)acc[i], false);
        if (OP == 1) acc[i] = (uint32_t)(__mul24((int)a[i], (int)c) + (int)acc[i]);
        if (OP == 2) acc[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, acc[i]) + __builtin_bit_cast(short2v, a[i]), __builtin_bit_cast(short2v, c)));
        if (OP == 3) acc[i] = __builtin_amdgcn_alignbit(acc[i], a[i], 16) + c;
        if (OP == 4) acc[i] = (acc[i] + a[i]) ^ c;
        if (OP == 5) acc[i] = __builtin_amdgcn_perm(acc[i], a[i], c);
      }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char* name, int ops_per_iter, uint32_t* in, uint32_t* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 8;          // 8 blocks/CU -> 8 waves/SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)blocks * 4 * iters * 32 * ops_per_iter;     // wave-instructions
  printf("%-28s %.3f ms  %.1f G wave-instr/s  = %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, ms, winstr / ms / 1e6,
         1024 * 2.4e9 / (winstr / (ms * 1e-3)));
}
int main() {
  uint32_t *in, *out; hipMalloc(&in, 8192); hipMalloc(&out, 256 * 8 * 256 * 4); hipMemset(in, 3, 8192);
  run<0>("v_dot2c_i32_i16", 1, in, out);
  run<1>("v_mad_i32_i24", 1, in, out);
  run<2>("v_pk_add_i16+v_pk_min_i16", 2, in, out);
  run<3>("v_alignbit+v_add", 2, in, out);
  run<4>("v_add+v_xor", 2, in, out);
  run<5>("v_perm_b32", 1, in, out);
  return 0;
}
