// Probe: exact lane/element semantics of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int l = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = l * 8;                                   // lane l -> halfwords 4l..4l+3
  else if (mode == 1) addr = (l & 15) * 128 + (l >> 4) * 8;      // 16 rows of 128 B, 8-B column block per group
  else addr = ((l & 3) * 16 + (l >> 2)) * 8;                      // another pattern
  addr += (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t*)lds;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("L%02d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf(l % 4 == 3 ? "\n" : "  |  "); }
  }
  return 0;
}
