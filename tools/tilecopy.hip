// micro-benchmark: what HBM gives a kernel that moves 2-D tiles of pitch-linear int16 planes (the access shape of the picture
// kernels) compared with a linear copy.  16 "pictures" of 3840x2160 int16 (pitch 7936 B as in libhmgpu), each block copies a
// TW x 64 sample tile src -> dst (different buffers), tiles in raster order, pictures interleaved as the MC kernels do.
// usage: tilecopy.bin        prints GB/s (read + write) per tile width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int W = 3840, H = 2160, PITCH = 3968, NP = 16;     // samples
template <int TWB>   // tile width in bytes (128, 256, 512, ...), tile height 64 rows; 256 threads, 16 B per thread per access
__global__ void __launch_bounds__(256) k_tile(const int16_t* const* src, int16_t* const* dst, int tiles_x, int dy, int dx) {
  const int pic = blockIdx.x, tile = blockIdx.y;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const char* s = (const char*)src[pic] + (size_t)(ty * 64 + dy) * PITCH * 2 + (size_t)tx * TWB + dx * 2;
  char* d = (char*)dst[pic] + (size_t)(ty * 64) * PITCH * 2 + (size_t)tx * TWB;
  constexpr int LPR = TWB / 16;            // lanes per row
  constexpr int RPI = 256 / LPR;           // rows per iteration
  const int lx = threadIdx.x % LPR, ly = threadIdx.x / LPR;
  u32x4 v[64 / RPI];
#pragma unroll
  for (int i = 0; i < 64 / RPI; i++) {
    const int y = i * RPI + ly;
    typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
    v[i] = (ty * 64 + y < H) ? *(const u32x4_a4*)(s + (size_t)y * PITCH * 2 + lx * 16) : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < 64 / RPI; i++) {
    const int y = i * RPI + ly;
    if (ty * 64 + y < H) *(u32x4*)(d + (size_t)y * PITCH * 2 + lx * 16) = v[i];
  }
}
__global__ void __launch_bounds__(256) k_linear(const u32x4* src, u32x4* dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
template <int TWB> double run(const int16_t* const* ds, int16_t* const* dd, int dy, int dx) {
  const int tiles_x = W * 2 / TWB, tiles_y = (H + 63) / 64;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 grid(NP, tiles_x * tiles_y);
  hipLaunchKernelGGL(k_tile<TWB>, grid, dim3(256), 0, 0, ds, dd, tiles_x, dy, dx);
  hipEventRecord(a);
  for (int it = 0; it < 10; it++) hipLaunchKernelGGL(k_tile<TWB>, grid, dim3(256), 0, 0, ds, dd, tiles_x, dy, dx);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return 10.0 * NP * 2.0 * W * H * 2 / (ms * 1e-3) / 1e9;
}
int main() {
  const size_t plane = (size_t)PITCH * (H + 160) * 2;
  int16_t *hs[NP], *hd[NP];
  for (int i = 0; i < NP; i++) { hipMalloc(&hs[i], plane); hipMalloc(&hd[i], plane); hipMemset(hs[i], i, plane); }
  int16_t **ds, **dd; hipMalloc(&ds, sizeof(hs)); hipMalloc(&dd, sizeof(hd));
  hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice); hipMemcpy(dd, hd, sizeof(hd), hipMemcpyHostToDevice);
  {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t n = plane / 16;
    hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, (const u32x4*)hs[0], (u32x4*)hd[0], n);
    hipEventRecord(a);
    for (int i = 0; i < NP; i++) hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, (const u32x4*)hs[i], (u32x4*)hd[i], n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("linear copy                 %.0f GB/s\n", NP * 2.0 * plane / (ms * 1e-3) / 1e9);
  }
  for (int dxy = 0; dxy < 2; dxy++) {
    const int dy = dxy ? 37 : 0, dx = dxy ? 22 : 0;   // displaced source (a motion vector): unaligned rows / lines
    printf("tile 128 B x 64 (dy %d dx %d)   %.0f GB/s\n", dy, dx, run<128>(ds, dd, dy, dx));
    printf("tile 256 B x 64 (dy %d dx %d)   %.0f GB/s\n", dy, dx, run<256>(ds, dd, dy, dx));
    printf("tile 512 B x 64 (dy %d dx %d)   %.0f GB/s\n", dy, dx, run<512>(ds, dd, dy, dx));
    printf("tile 1024 B x 64 (dy %d dx %d)  %.0f GB/s\n", dy, dx, run<1024>(ds, dd, dy, dx));
  }
  return 0;
}
