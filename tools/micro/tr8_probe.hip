// what does ds_read_b64_tr_b8 return?  LDS image: 64 rows x 64 bytes; pass 0: byte = row, pass 1: byte = column.
// lane l reads at address (l / 8) * 64 + (l % 8) * 8 (mode 0: lane-linear 8-byte chunks) or, mode 1, row = l % 16, chunk = l / 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
__global__ void k(uint8_t* o, int pass, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = pass ? (uint8_t)(i % 64) : (uint8_t)(i / 64);
  __syncthreads();
  const int l = threadIdx.x;
  const int addr = mode == 0 ? (l / 8) * 64 + (l % 8) * 8 : (l % 16) * 64 + (l / 16) * 8;
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((i32x2 __attribute__((address_space(3)))*)(lds + addr));
  *(i32x2*)(o + l * 8) = v;
}
int main() {
  uint8_t *d, h[2][512];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 2; ++mode) {
    for (int p = 0; p < 2; ++p) { k<<<1, 64>>>(d, p, mode); hipMemcpy(h[p], d, 512, hipMemcpyDeviceToHost); }
    printf("mode %d: lane -> 8 x (row,col)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 8; ++j) printf(" (%2d,%2d)", h[0][l * 8 + j], h[1][l * 8 + j]);
      printf("\n");
    }
  }
  return 0;
}
