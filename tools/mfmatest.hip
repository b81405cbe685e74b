// mfmatest.hip -- the review's MFMA question for the MC H pass, answered on the device:
//   (1) is  In[16 x 32] x T[32 x 16]  on v_mfma_f32_16x16x32_f16 EXACT for 10-bit samples and HM's taps?  Samples enter as f16 either by
//       OR-ing 0x6400 (1024 + x, one v_or per dword) or raw (the int16 bits read as f16 denormals, x * 2^-24: no instruction at all);
//   (2) what does an H step cost that way (two tiles x 16 window rows per MFMA, block-diagonal Toeplitz taps, results converted to the
//       packed 15-bit row pairs the V pass reads) against the v_dot2 form of k_mc.hip (two rows x 8 columns per lane)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfmatest.bin tools/mfmatest.hip ; run on the GPU box: ./tools/mfmatest.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_exact(const unsigned short* __restrict__ x, const short* __restrict__ t, float* __restrict__ out, int mode) {
  // x: [16][32] samples, t: [32][16] taps; lane: row / col = lane % 16, k = (lane / 16) * 8 + j
  const int lane = threadIdx.x, rc = lane & 15, kb = (lane >> 4) * 8;
  unsigned short a[8];
  _Float16 b[8];
  for (int j = 0; j < 8; j++) { a[j] = x[rc * 32 + kb + j]; b[j] = (_Float16)(float)t[(kb + j) * 16 + rc]; }
  half8 av, bv;
  for (int j = 0; j < 8; j++) {
    const unsigned short bits = mode == 0 ? (unsigned short)(a[j] | 0x6400) : a[j];     // 1024 + x, or the denormal x * 2^-24
    av[j] = __builtin_bit_cast(_Float16, bits);
    bv[j] = b[j];
  }
  float4v c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
  for (int i = 0; i < 4; i++) out[(4 * (lane >> 4) + i) * 16 + rc] = c[i];
}

// timing: per iteration one H step of two tiles x 16 rows.  MFMA form: operand from registers (as after a 16-byte window load), 4 v_or,
// MFMA, 4 magic adds (shift + floor for free), 2 packs, one 8-byte LDS store.  dot2 form: the lane's share of the same work = its body
// item of k_mc.hip (2 rows x 8 columns: 80 v_dot2 + 8 packs) -- 64 lanes x 16 samples = two tiles x 16 rows x 8 columns x 2... per wave both
// forms produce 2 x 16 x 8 = 256 intermediates per (MFMA step) vs 64 x 16 = 1024 per (dot2 item round): cycles are reported per intermediate.
__global__ void k_time_mfma(unsigned* __restrict__ sink, int iters, unsigned long long* __restrict__ cycles) {
  __shared__ unsigned lds[64 * 2 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 w = {(unsigned)lane * 2654435761u & 0x03ff03ffu, (unsigned)(lane + 7) * 40503u & 0x03ff03ffu, (unsigned)lane * 97u & 0x03ff03ffu, (unsigned)lane * 31u & 0x03ff03ffu};
  half8 bv;
  for (int j = 0; j < 8; j++) bv[j] = (_Float16)(float)((lane + j) % 7 - 3);
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
    u32x4 o = {w.x | 0x64006400u, w.y | 0x64006400u, w.z | 0x64006400u, w.w | 0x64006400u};
    float4v c = {-65536.f * 0.25f, -65536.f * 0.25f, -65536.f * 0.25f, -65536.f * 0.25f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, o), bv, c, 0, 0, 0);
    const float M = 12582912.f;              // 1.5 * 2^23: the integer lands in the mantissa
    const unsigned i0 = __builtin_bit_cast(unsigned, c[0] + M), i1 = __builtin_bit_cast(unsigned, c[1] + M), i2 = __builtin_bit_cast(unsigned, c[2] + M), i3 = __builtin_bit_cast(unsigned, c[3] + M);
    const unsigned p0 = __builtin_amdgcn_perm(i1, i0, 0x05040100u), p1 = __builtin_amdgcn_perm(i3, i2, 0x05040100u);
    lds[(wave * 64 + lane) * 2] = p0; lds[(wave * 64 + lane) * 2 + 1] = p1;
    w.x += p0 & 0x00010001u; acc += p1;        // (a dependency, so that nothing is hoisted)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[lane];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
__device__ inline int dot2(unsigned a, unsigned b, int c) { int d; asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__global__ void k_time_dot2(unsigned* __restrict__ sink, int iters, unsigned long long* __restrict__ cycles) {
  __shared__ unsigned lds[64 * 4 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned d[2][8];
  for (int r = 0; r < 2; r++) for (int i = 0; i < 8; i++) d[r][i] = (unsigned)(lane * 131 + r * 17 + i * 7) & 0x03ff03ffu;
  unsigned te[5], to[5];
  for (int j = 0; j < 5; j++) { te[j] = 0x0004ffffu + j; to[j] = 0x003afff6u + j; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
    int sum[2][8];
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int x = 0; x < 8; x++) {
        int v = dot2(d[r][x >> 1], (x & 1) ? to[0] : te[0], 0);
#pragma unroll
        for (int j = 1; j < 5; j++) v = dot2(d[r][((x >> 1) + j) & 7], (x & 1) ? to[j] : te[j], v);
        sum[r][x] = v;
      }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const unsigned p0 = __builtin_amdgcn_perm((unsigned)(sum[1][2 * c] >> 2), (unsigned)(sum[0][2 * c] >> 2), 0x05040100u);
      const unsigned p1 = __builtin_amdgcn_perm((unsigned)(sum[1][2 * c + 1] >> 2), (unsigned)(sum[0][2 * c + 1] >> 2), 0x05040100u);
      lds[((wave * 64 + lane) * 4 + c) * 2] = p0; lds[((wave * 64 + lane) * 4 + c) * 2 + 1] = p1;
      acc += p0 ^ p1;
    }
    d[0][0] += acc & 1u;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[lane];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  const int luma[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
  unsigned short* dx; short* dt; float* dout;
  hipMalloc(&dx, 16 * 32 * 2); hipMalloc(&dt, 32 * 16 * 2); hipMalloc(&dout, 256 * 4);
  long bad[2] = {0, 0}, total = 0;
  srand(1);
  for (int trial = 0; trial < 2000; trial++) {
    std::vector<unsigned short> x(16 * 32);
    std::vector<short> t(32 * 16, 0);
    for (auto& v : x) v = trial < 8 ? (trial & 1 ? 1023 : 0) : (unsigned short)(rand() & 1023);
    // block-diagonal Toeplitz: tile A = k 0..15 / columns 0..7, tile B = k 16..31 / columns 8..15; phase and parity per tile
    for (int tile = 0; tile < 2; tile++) {
      const int f = rand() & 3, par = rand() & 1;
      for (int n = 0; n < 8; n++) for (int j = 0; j < 8; j++) { const int k = n + j + par; if (k < 16) t[(tile * 16 + k) * 16 + tile * 8 + n] = (short)luma[f][j]; }
    }
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dt, t.data(), t.size() * 2, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
      hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dx, dt, dout, mode);
      float out[256];
      hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
      for (int m = 0; m < 16; m++)
        for (int n = 0; n < 16; n++) {
          long s = 0, ts = 0;
          for (int k = 0; k < 32; k++) { s += (long)x[m * 32 + k] * t[k * 16 + n]; ts += t[k * 16 + n]; }
          const double want = mode == 0 ? (double)(s + 1024 * ts) : (double)s / 16777216.0;
          if ((double)out[m * 16 + n] != want) bad[mode]++;
          total++;
        }
    }
  }
  printf("exactness: %ld sums per form; wrong: OR-0x6400 form %ld, denormal form %ld\n", total / 2, bad[0], bad[1]);
  unsigned* sink; unsigned long long* cyc;
  hipMalloc(&sink, (size_t)2048 * 256 * 4); hipMalloc(&cyc, 8);   // the largest launch below: 2048 blocks of 256 threads
  const int iters = 2000;
  for (int waves = 1; waves <= 8; waves *= 2) {
    unsigned long long c1 = 0, c2 = 0;
    // one workgroup per CU-ish: 256 blocks x (waves x 64) threads keeps `waves`/4.. per SIMD; report the cycles of block 0
    hipLaunchKernelGGL(k_time_mfma, dim3(1024), dim3(64 * (waves > 4 ? 4 : waves)), 0, 0, sink, iters, cyc); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_time_mfma, dim3(waves > 4 ? 2048 : 1024), dim3(64 * (waves > 4 ? 4 : waves)), 0, 0, sink, iters, cyc); hipMemcpy(&c1, cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_time_dot2, dim3(waves > 4 ? 2048 : 1024), dim3(64 * (waves > 4 ? 4 : waves)), 0, 0, sink, iters, cyc); hipMemcpy(&c2, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d waves per workgroup: MFMA step %.1f cycles (256 intermediates: %.3f cycles each), dot2 item round %.1f cycles (1024 intermediates: %.3f each)\n",
           waves > 4 ? 4 : waves, (double)c1 / iters, (double)c1 / iters / 256, (double)c2 / iters, (double)c2 / iters / 1024);
  }
  return (bad[0] || bad[1]) ? 1 : 0;
}
