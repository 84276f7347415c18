// Instruction-rate probes for gfx950 (one-off measurement tool, not part of the product path).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ubench tools/probes/ubench.hip && tools/probes/ubench
// For each probe: a kernel whose waves run REP iterations of a straight-line block of N instructions of one kind
// (independent chains unless noted), W waves per SIMD, every CU busy. Reports shader cycles (s_memtime) per instruction
// per wave and per SIMD, so the per-tile cycle budgets of the attention kernels can be priced.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define REP 256

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int rep) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (float)(threadIdx.x & 7)); fb[i] = (__bf16)0.5f; }
  const float c = 1.0001f, d = 1e-4f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rep; ++r) {
    if (KIND == 0) {  // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
    } else if (KIND == 1) {  // 16 independent v_exp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 2) {  // 8 independent v_pk_fma_f32 (2 lanes-elements each)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef __attribute__((ext_vector_type(2))) float f2;
        f2 x = {a[2 * i], a[2 * i + 1]}, cc = {c, c}, dd = {d, d};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(cc), "v"(dd));
        a[2 * i] = x[0]; a[2 * i + 1] = x[1];
      }
    } else if (KIND == 3) {  // 8 independent MFMA 16x16x32 bf16
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
    } else if (KIND == 4) {  // 8 MFMA, each followed by 4 independent v_fma (do they overlap inside ONE wave?)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2 * i]) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2 * i + 1]) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(2 * i + 2) & 15]) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(2 * i + 3) & 15]) : "v"(c), "v"(d));
      }
    } else if (KIND == 5) {  // 8 MFMA, each followed by 2 v_exp
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * i + 1]));
      }
    } else if (KIND == 6) {  // 16 v_cvt_pk_bf16_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        uint32_t o;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(a[i]), "v"(a[(i + 1) & 15]));
        a[i] = __uint_as_float(o & 0x3f800000u);
      }
    } else if (KIND == 7) {  // phase-alternating wave: 16 MFMA then 32 (fma + exp) that depend on them, then 16 MFMA ...
      // (the attention tile skeleton: do several waves per SIMD overlap each other's phases?)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[i], 0, 0, 0);
      float e[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float x = acc[i >> 2][i & 3];
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[i]) : "v"(x), "v"(c), "v"(d));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t p0, p1, p2, p3;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p0) : "v"(e[4 * i]), "v"(e[4 * i + 1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(e[4 * i + 2]), "v"(e[4 * i + 3]));
        p2 = p0; p3 = p1;
        typedef __attribute__((ext_vector_type(4))) uint32_t u4;
        u4 pk = {p0, p1, p2, p3};
        fb = __builtin_bit_cast(bf16x8_t, pk);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        acc[(i + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[(i + 4) & 7], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_inst, float* out, unsigned long long* cyc, int ncu) {
  for (int bpc = 1; bpc <= 4; ++bpc) {  // blocks of 4 waves per CU = waves per SIMD
    const int grid = ncu * bpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<grid, 256>>>(out, cyc, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<grid, 256>>>(out, cyc, REP);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= grid;
    const double per_wave = mean / (double)(REP * n_inst);
    printf("%-34s waves/SIMD %d: %8.2f cyc/inst/wave  %7.2f cyc/inst/SIMD   kernel %8.1f us (%.2f GHz equiv)\n", name, bpc, per_wave,
           per_wave / bpc, ms * 1e3, mean / (ms * 1e-3) / 1e9);
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("%s: %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)ncu * 4 * 256 * 4);
  hipMalloc(&cyc, (size_t)ncu * 4 * 8);
  run<0>("v_fma_f32 x16 (indep)", 16, out, cyc, ncu);
  run<1>("v_exp_f32 x16 (indep)", 16, out, cyc, ncu);
  run<2>("v_pk_fma_f32 x8 (indep)", 8, out, cyc, ncu);
  run<6>("v_cvt_pk_bf16_f32 x16", 16, out, cyc, ncu);
  run<3>("mfma16x16x32 x8 (indep)", 8, out, cyc, ncu);
  run<4>("8 x (mfma + 4 v_fma)  [per group]", 8, out, cyc, ncu);
  run<5>("8 x (mfma + 2 v_exp)  [per group]", 8, out, cyc, ncu);
  run<7>("attention skeleton [per tile]", 1, out, cyc, ncu);
  return 0;
}
