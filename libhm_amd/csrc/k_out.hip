// k_out.hip -- the output side of a finished picture: sample packing for the application and the
// decoded-picture-hash check without moving the picture.
//   TVideoIOYuv::write / writePlane (TVideoIOYuv.cpp:362-480, 706-790): 8- or 16-bit samples, cropped to the conformance window
//   compCRC / compChecksum (TComPicYuvMD5.cpp:87-125, 139-163)
#include "hmgpu_dev.h"

namespace hmgpu {

// ---- packing: one thread per output sample pair (bytes = 1) or per sample pair (bytes = 2) of the cropped plane
__global__ void __launch_bounds__(256) k_pack(const int16_t* __restrict__ src, int pitch, int x0, int y0, int w, int h, int bytes,
                                              uint8_t* __restrict__ dst, int dst_stride) {
  const int x = (blockIdx.x * 256 + threadIdx.x) * 2, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const int16_t* s = src + (ptrdiff_t)(y0 + y) * pitch + x0 + x;
  const int a = (uint16_t)ldg(s), b = x + 1 < w ? (uint16_t)ldg(s + 1) : 0;
  uint8_t* d = dst + (size_t)y * dst_stride + (size_t)x * bytes;
  if (bytes == 1) { d[0] = (uint8_t)a; if (x + 1 < w) d[1] = (uint8_t)b; }
  else { d[0] = (uint8_t)a; d[1] = (uint8_t)(a >> 8); if (x + 1 < w) { d[2] = (uint8_t)b; d[3] = (uint8_t)(b >> 8); } }
}

void launch_pack(const int16_t* src, int pitch, int x0, int y0, int w, int h, int bytes, uint8_t* dst, int dst_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((w / 2 + 256) / 256), (unsigned)h), dim3(256), 0, s, src, pitch, x0, y0, w, h, bytes, dst, dst_stride);
}

// ---- checksum: sum over the plane of (byte ^ mask(x, y)) mod 2^32 -- any order
__global__ void __launch_bounds__(256) k_checksum(const int16_t* __restrict__ src, int pitch, int w, int h, int bd, uint32_t* __restrict__ out) {
  __shared__ uint32_t part[256];
  uint32_t sum = 0;
  for (int y = blockIdx.x; y < h; y += gridDim.x)
    for (int x = threadIdx.x; x < w; x += 256) {
      const uint32_t mask = ((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)) & 0xff, v = (uint16_t)ldg(src + (ptrdiff_t)y * pitch + x);
      sum += (v & 0xff) ^ mask;
      if (bd > 8) sum += (v >> 8) ^ mask;
    }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(out, part[0]);
}

// ---- CRC-16 (polynomial 0x1021, MSB first).  The CRC register after a message is a linear function of (register before,
// message): crc(s, A || B) = shift(crc(s, A), |B|) ^ crc(0, B), where shift multiplies by x^(8|B|) modulo the polynomial.
// Stage 1: one thread per row computes crc(0, row).  Stage 2: one thread folds the rows in order, starting from HM's 0xffff.
__device__ inline uint32_t crc_byte(uint32_t crc, uint32_t byte) {
#pragma unroll
  for (int b = 7; b >= 0; b--) {
    const uint32_t msb = (crc >> 15) & 1;
    crc = (((crc << 1) | ((byte >> b) & 1)) & 0xffffu) ^ (msb * 0x1021u);
  }
  return crc;
}
__global__ void __launch_bounds__(256) k_crc_rows(const int16_t* __restrict__ src, int pitch, int w, int h, int bd, uint32_t* __restrict__ rows) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= h) return;
  uint32_t crc = 0;
  const int16_t* s = src + (ptrdiff_t)y * pitch;
  for (int x = 0; x < w; x++) {
    const uint32_t v = (uint16_t)ldg(s + x);
    crc = crc_byte(crc, v & 0xff);                          // HM feeds the low byte first, then the high byte (:95-115)
    if (bd > 8) crc = crc_byte(crc, v >> 8);
  }
  rows[y] = crc;
}
__global__ void k_crc_fold(const uint32_t* __restrict__ rows, int h, int row_bytes, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // the register after shifting in row_bytes zero bytes, as a 16x16 bit matrix: column k = image of bit k
  uint32_t col[16];
  for (int k = 0; k < 16; k++) {
    uint32_t c = 1u << k;
    for (int i = 0; i < row_bytes; i++) c = crc_byte(c, 0);
    col[k] = c;
  }
  uint32_t crc = 0xffff;
  for (int y = 0; y < h; y++) {
    uint32_t s = 0;
    for (int k = 0; k < 16; k++) if ((crc >> k) & 1) s ^= col[k];
    crc = s ^ rows[y];
  }
  for (int b = 0; b < 16; b++) { const uint32_t msb = (crc >> 15) & 1; crc = ((crc << 1) & 0xffffu) ^ (msb * 0x1021u); }
  *out = crc;
}

void launch_checksum(const int16_t* src, int pitch, int w, int h, int bd, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_checksum, dim3((unsigned)((h + 3) / 4)), dim3(256), 0, s, src, pitch, w, h, bd, out);
}
void launch_crc(const int16_t* src, int pitch, int w, int h, int bd, uint32_t* rows, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_crc_rows, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, s, src, pitch, w, h, bd, rows);
  hipLaunchKernelGGL(k_crc_fold, dim3(1), dim3(64), 0, s, rows, h, w * (bd > 8 ? 2 : 1), out);
}

}  // namespace hmgpu
