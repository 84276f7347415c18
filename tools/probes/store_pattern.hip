// Store-pattern probe (round 6): does the SHAPE of the epilogue's 16-byte stores matter? The 256 x 256 kernels store a wave's
// 128 x 64 bf16 output as instructions of 16 rows x 64 B (4 lanes per row segment: half a 128-B line, the other half one
// instruction later). Pattern B writes 8 rows x 128 B (8 lanes per row: whole lines) per instruction. Same bytes, same tiles,
// one block per CU, every block storing its tiles back to back (the burst of a lock-step epilogue), nothing else running.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/store_pattern tools/probes/store_pattern.hip && tools/probes/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(uint4* out, int ld_bytes, int tiles_per_block, int tiles_c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const uint4 v = make_uint4(tid, blockIdx.x, 3, 4);
  for (int t = 0; t < tiles_per_block; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int tr = tile / tiles_c, tc = tile % tiles_c;
    char* base = (char*)out + (size_t)(tr * 256 + wr * 128) * ld_bytes + (size_t)(tc * 256 + wc * 64) * 2;
    if (PATTERN == 0) {  // the kernels' pattern: lane (l15, g): row fm*16 + l15, 16-B chunk g (+4 for q = 1)
      const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            *(uint4*)(base + (size_t)(mq * 64 + fm * 16 + l15) * ld_bytes + (g + 4 * q) * 16) = v;
    } else {  // whole lines: lane (r8, c8): row i*8 + r8, chunk c8
      const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
      for (int i = 0; i < 16; ++i) *(uint4*)(base + (size_t)(i * 8 + r8) * ld_bytes + c8 * 16) = v;
    }
  }
}
int main() {
  const int M = 8192, N = 9728;  // gate|up output
  const size_t bytes = (size_t)M * N * 2;
  uint4* d;
  hipMalloc(&d, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles_c = N / 256, tiles = (M / 256) * tiles_c, blocks = 256, tpb = tiles / blocks;  // 1216 tiles -> 4 per block (+ remainder ignored)
  for (int rep = 0; rep < 3; ++rep)
    for (int pat = 0; pat < 2; ++pat) {
      float best = 1e9f;
      for (int it = 0; it < 20; ++it) {
        hipEventRecord(e0);
        if (pat == 0) store_kernel<0><<<blocks, 512>>>(d, N * 2, tpb, tiles_c);
        else store_kernel<1><<<blocks, 512>>>(d, N * 2, tpb, tiles_c);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double gb = (double)blocks * tpb * 256 * 256 * 2 / 1e9;
      printf("pattern %s: %.1f us for %.1f MB -> %.2f TB/s\n", pat == 0 ? "A 16 rows x 64 B " : "B 8 rows x 128 B ", best * 1e3, gb * 1e3, gb / best);
    }
  return 0;
}
