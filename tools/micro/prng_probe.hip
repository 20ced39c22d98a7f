// v_prng_b32 (gfx950): what does one step do to its source, and are thresholded halves of successive outputs usable as dropout decisions?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* o, int steps) {
  unsigned s = (threadIdx.x + blockIdx.x * 256) * 2654435761u + 0x1234567u;
  for (int i = 0; i < steps; ++i) { s = __builtin_amdgcn_prng_b32(s); o[(blockIdx.x * 256 + threadIdx.x) * steps + i] = s; }
}
int main() {
  const int steps = 64, n = 256 * 64;
  unsigned* d; hipMalloc(&d, n * steps * 4);
  k<<<64, 256>>>(d, steps);
  std::vector<unsigned> h(n * steps);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  printf("lane 0:"); for (int i = 0; i < 8; ++i) printf(" %08x", h[i]); printf("\n");
  printf("lane 1:"); for (int i = 0; i < 8; ++i) printf(" %08x", h[steps + i]); printf("\n");
  const unsigned thr = 6554;  // p = 0.1 on 16 bits
  double keep = 0, both = 0, succ = 0, cnt = 0, cnts = 0;
  for (int t = 0; t < n; ++t)
    for (int i = 0; i < steps; ++i) {
      const unsigned v = h[t * steps + i];
      const int k0 = (v & 0xFFFF) >= thr, k1 = (v >> 16) >= thr;
      keep += k0 + k1; both += (!k0 && !k1); cnt += 1;
      if (i + 1 < steps) { const unsigned w = h[t * steps + i + 1]; succ += (!k0 && !((w & 0xFFFF) >= thr)); cnts += 1; }
    }
  printf("keep rate %.5f (expect 0.90000); P(both halves dropped) %.5f (expect 0.01000); P(low half dropped in two successive outputs) %.5f (expect 0.01000)\n",
         keep / (2 * cnt), both / cnt, succ / cnts);
  return 0;
}
