// does v_pk_mad_i16 with clamp saturate the exact product?  and v_cvt_pk_i16_i32, v_pk_add_i16 clamp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* in, uint32_t* out) {
  uint32_t q = in[threadIdx.x], s = in[64 + threadIdx.x], d, e;
  asm("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(q), "v"(s));
  asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(e) : "v"(q), "v"(s));
  out[threadIdx.x] = d; out[64 + threadIdx.x] = e;
  int a = (int)in[128 + threadIdx.x], b = (int)in[192 + threadIdx.x];
  uint32_t c;
  asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b));
  out[128 + threadIdx.x] = c;
}
static int sat(long long v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int)v); }
int main() {
  uint32_t h[256], r[192];
  for (int i = 0; i < 64; i++) {
    int16_t q0 = (int16_t)(i * 1031 - 30000), q1 = (int16_t)(32767 - i * 997), s0 = (int16_t)(40 << (i % 9)), s1 = (int16_t)(72 << (i % 8));
    h[i] = (uint16_t)q0 | ((uint32_t)(uint16_t)q1 << 16); h[64 + i] = (uint16_t)s0 | ((uint32_t)(uint16_t)s1 << 16);
    h[128 + i] = (uint32_t)(i * 2999 - 90000); h[192 + i] = (uint32_t)(70000 - i * 2500);
  }
  uint32_t *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(r));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) {
    int16_t q0 = (int16_t)(h[i] & 0xffff), q1 = (int16_t)(h[i] >> 16), s0 = (int16_t)(h[64 + i] & 0xffff), s1 = (int16_t)(h[64 + i] >> 16);
    int w0 = sat((long long)q0 * s0), w1 = sat((long long)q1 * s1);
    int g0 = (int16_t)(r[i] & 0xffff), g1 = (int16_t)(r[i] >> 16);
    if (g0 != w0 || g1 != w1) { if (bad < 5) printf("mad i=%d q=(%d,%d) s=(%d,%d) got (%d,%d) want (%d,%d)\n", i, q0, q1, s0, s1, g0, g1, w0, w1); bad++; }
    int a0 = sat((long long)q0 + s0), a1 = sat((long long)q1 + s1);
    if ((int16_t)(r[64 + i] & 0xffff) != a0 || (int16_t)(r[64 + i] >> 16) != a1) { if (bad < 5) printf("add i=%d\n", i); bad++; }
    int c0 = sat((int)h[128 + i]), c1 = sat((int)h[192 + i]);
    if ((int16_t)(r[128 + i] & 0xffff) != c0 || (int16_t)(r[128 + i] >> 16) != c1) { if (bad < 5) printf("cvt i=%d got (%d,%d) want (%d,%d)\n", i, (int16_t)(r[128+i]&0xffff), (int16_t)(r[128+i]>>16), c0, c1); bad++; }
  }
  printf("mismatches: %d\n", bad);
  return 0;
}
