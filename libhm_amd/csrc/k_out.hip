// k_out.hip -- the output side of a finished picture: sample packing for the application and the
// decoded-picture-hash check without moving the picture.
//   TVideoIOYuv::write / writePlane (TVideoIOYuv.cpp:362-480, 706-790): 8- or 16-bit samples, cropped to the conformance window
//   compCRC / compChecksum (TComPicYuvMD5.cpp:87-125, 139-163)
#include "hmgpu_dev.h"

namespace hmgpu {

// ---- packing: one thread per output sample pair (bytes = 1) or per sample pair (bytes = 2) of the cropped plane
// step: distance of two samples of the plane in memory (1: luma; 2: a chroma component, whose samples alternate with the other one's:
// hmgpu_dev.h "chroma planes")
__global__ void __launch_bounds__(256) k_pack(const int16_t* __restrict__ src, int pitch, int step, int x0, int y0, int w, int h, int bytes,
                                              uint8_t* __restrict__ dst, int dst_stride) {
  const int x = (blockIdx.x * 256 + threadIdx.x) * 2, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const int16_t* s = src + (ptrdiff_t)(y0 + y) * pitch + (x0 + x) * step;
  const int a = (uint16_t)ldg(s), b = x + 1 < w ? (uint16_t)ldg(s + step) : 0;
  uint8_t* d = dst + (size_t)y * dst_stride + (size_t)x * bytes;
  if (bytes == 1) { d[0] = (uint8_t)a; if (x + 1 < w) d[1] = (uint8_t)b; }
  else { d[0] = (uint8_t)a; d[1] = (uint8_t)(a >> 8); if (x + 1 < w) { d[2] = (uint8_t)b; d[3] = (uint8_t)(b >> 8); } }
}

void launch_pack(const int16_t* src, int pitch, int step, int x0, int y0, int w, int h, int bytes, uint8_t* dst, int dst_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((w / 2 + 256) / 256), (unsigned)h), dim3(256), 0, s, src, pitch, step, x0, y0, w, h, bytes, dst, dst_stride);
}

// ---- the way in (hmgpu_picture_upload): a dense w x h block of samples into a plane whose samples lie `step` apart
__global__ void __launch_bounds__(256) k_unpack(const int16_t* __restrict__ src, int w, int h, int16_t* __restrict__ dst, int pitch, int step) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  stg(dst + (ptrdiff_t)y * pitch + x * step, ldg(src + (size_t)y * w + x));
}
void launch_unpack(const int16_t* src, int w, int h, int16_t* dst, int pitch, int step, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((w + 255) / 256), (unsigned)h), dim3(256), 0, s, src, w, h, dst, pitch, step);
}

// ---- checksum: sum over the plane of (byte ^ mask(x, y)) mod 2^32 -- any order
__global__ void __launch_bounds__(256) k_checksum(const int16_t* __restrict__ src, int pitch, int step, int w, int h, int bd, uint32_t* __restrict__ out) {
  __shared__ uint32_t part[256];
  uint32_t sum = 0;
  for (int y = blockIdx.x; y < h; y += gridDim.x)
    for (int x = threadIdx.x; x < w; x += 256) {
      const uint32_t mask = ((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)) & 0xff, v = (uint16_t)ldg(src + (ptrdiff_t)y * pitch + x * step);
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
__global__ void __launch_bounds__(256) k_crc_rows(const int16_t* __restrict__ src, int pitch, int step, int w, int h, int bd, uint32_t* __restrict__ rows) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= h) return;
  uint32_t crc = 0;
  const int16_t* s = src + (ptrdiff_t)y * pitch;
  for (int x = 0; x < w; x++) {
    const uint32_t v = (uint16_t)ldg(s + x * step);
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

// ---- MD5 (RFC 1321; HM: libmd5/libmd5.c through TComPicYuvMD5.cpp:183-205) of the packed planes, the hash every HM-encoded stream
// carries by default.  The digest of a plane is ONE serial chain of 64-byte blocks (~26 cycles per step, 64 steps per block: ~0.2 s
// for a 3840x2160 10-bit luma plane, eight times what a host core takes), and nothing inside a chain is parallel.  What the device
// offers is chains side by side: every LANE of a wave runs the chain of another plane -- same instruction stream, its own message
// (64 bytes per block straight from its plane, one block ahead) and state -- so one wave hashes the planes of 21 pictures in the
// time of the longest of them (a batch is up to two such waves).  hmgpu_picture_hash_begin collects the planes of finished pictures and launches them in batches on
// low-priority streams of their own.  Plane sizes are multiples of 16 bytes (width and height are multiples of 8).
__device__ inline uint32_t md5_rol(uint32_t v, int s) { return __builtin_amdgcn_alignbit(v, v, 32 - s); }
__device__ inline void md5_block(const uint32_t (&m)[16], uint32_t& sa, uint32_t& sb, uint32_t& sc, uint32_t& sd) {
  constexpr uint32_t K[64] = {
      0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
      0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
      0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
      0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
      0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
      0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
  constexpr int S[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
  uint32_t a = sa, b = sb, c = sc, d = sd;
#pragma unroll
  for (int i = 0; i < 64; i++) {
    const int r = i >> 4;
    const int g = r == 0 ? i : r == 1 ? (5 * i + 1) & 15 : r == 2 ? (3 * i + 5) & 15 : (7 * i) & 15;
    const uint32_t f = r == 0 ? ((b & c) | (~b & d)) : r == 1 ? ((d & b) | (~d & c)) : r == 2 ? (b ^ c ^ d) : (c ^ (b | ~d));
    const uint32_t t = d;
    d = c; c = b;
    b = b + md5_rol(a + f + (m[g] + K[i]), S[r][i & 3]);
    a = t;
  }
  sa += a; sb += b; sc += c; sd += d;
}
__device__ inline void md5_fetch(const uint32_t* p, uint32_t (&m)[16]) {
#pragma unroll
  for (int k = 0; k < 4; k++) { const u32x4 v = ldg4(p + 4 * k); m[4 * k] = v.x; m[4 * k + 1] = v.y; m[4 * k + 2] = v.z; m[4 * k + 3] = v.w; }
}
__global__ void __launch_bounds__(64) k_md5(Md5Batch job) {
  const int lane = blockIdx.x * 64 + threadIdx.x;
  const bool live = lane < job.n;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(live ? job.msg[lane] : job.msg[0]);
  const unsigned long long n = live ? job.bytes[lane] : 0ull, blocks = n >> 6;
  uint32_t sa = 0x67452301u, sb = 0xefcdab89u, sc = 0x98badcfeu, sd = 0x10325476u;
  // four blocks per round, the next four in flight meanwhile (a lane's loads are its own, 48 planes = 48 places: ~3 us of chain per
  // round cover their latency; one wave per SIMD, registers are free)
  constexpr int R = 4;
  uint32_t cur[R][16], nxt[R][16];
#pragma unroll
  for (int j = 0; j < R; j++)
#pragma unroll
    for (int k = 0; k < 16; k++) cur[j][k] = nxt[j][k] = 0;
#pragma unroll
  for (int j = 0; j < R; j++) if ((unsigned long long)j < blocks) md5_fetch(p + j * 16, cur[j]);
  for (unsigned long long b = 0; __any(b < blocks); b += R) {
#pragma unroll
    for (int j = 0; j < R; j++) if (b + R + j < blocks) md5_fetch(p + (b + R + j) * 16, nxt[j]);
#pragma unroll
    for (int j = 0; j < R; j++) if (b + j < blocks) md5_block(cur[j], sa, sb, sc, sd);
#pragma unroll
    for (int j = 0; j < R; j++)
#pragma unroll
      for (int k = 0; k < 16; k++) cur[j][k] = nxt[j][k];
  }
  if (!live) return;
  // the rest of the message (0, 16, 32 or 48 bytes), the 0x80 byte, zeros and the length in bits in the last eight bytes of the last block
  const uint32_t r = (uint32_t)(n & 63u), tail_blocks = r + 9u > 64u ? 2u : 1u;
  const unsigned long long bits = n * 8ull;
  for (uint32_t tb = 0; tb < tail_blocks; tb++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t idx = tb * 16u + k;
      uint32_t w = 0;
      if (4u * idx < r) w = ldg(p + blocks * 16 + idx);
      else if (4u * idx == r) w = 0x80u;
      if (idx == tail_blocks * 16u - 2u) w = (uint32_t)bits;
      if (idx == tail_blocks * 16u - 1u) w = (uint32_t)(bits >> 32);
      cur[0][k] = w;
    }
    md5_block(cur[0], sa, sb, sc, sd);
  }
  uint32_t* out = job.out[lane];
  out[0] = sa; out[1] = sb; out[2] = sc; out[3] = sd;
}
void launch_md5(const Md5Batch& job, hipStream_t s) {
  hipLaunchKernelGGL(k_md5, dim3((unsigned)((job.n + 63) / 64)), dim3(64), 0, s, job);
}

void launch_checksum(const int16_t* src, int pitch, int step, int w, int h, int bd, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_checksum, dim3((unsigned)((h + 3) / 4)), dim3(256), 0, s, src, pitch, step, w, h, bd, out);
}
void launch_crc(const int16_t* src, int pitch, int step, int w, int h, int bd, uint32_t* rows, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_crc_rows, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, s, src, pitch, step, w, h, bd, rows);
  hipLaunchKernelGGL(k_crc_fold, dim3(1), dim3(64), 0, s, rows, h, w * (bd > 8 ? 2 : 1), out);
}

}  // namespace hmgpu
