// rounding / saturation of v_cvt_pk_u8_f32 (candidate for the 8-bit GELU' pack): hipcc --offload-arch=gfx950 -o cvt_u8_probe cvt_u8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* o, int n) {
  int i = threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAABBCCDDu);
}
int main() {
  const float h[] = {0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.7f, 255.4f, 255.5f, 300.f, -0.4f, -3.f, 26.499f, 26.5f, 226.5f};
  const int n = sizeof(h) / sizeof(float);
  float* dx; unsigned* dout;
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dout, n);
  unsigned r[32]; hipMemcpy(r, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%9.3f -> byte %3u  word %08x\n", h[i], (r[i] >> 8) & 255, r[i]);
  return 0;
}
