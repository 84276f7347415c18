// Common device helpers for the gfx950 (CDNA4) SpeechLM engine.
// Wave = 64 lanes; all kernels here are written for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define SLAM_DEVICE __device__ __forceinline__

SLAM_DEVICE float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even: gfx950 has v_cvt_pk_bf16_f32 (one instruction per pair)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
SLAM_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
SLAM_DEVICE bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

SLAM_DEVICE void unpack_bf16x8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

SLAM_DEVICE uint4 pack_bf16x8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

// raw v_exp_f32 (2^x): no denormal-range fix-up code (arguments here are <= 0 or moderate)
SLAM_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// sigmoid via v_exp_f32 + v_rcp_f32 (about 1 ulp each): 4 VALU ops instead of an IEEE division sequence
SLAM_DEVICE float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + fast_exp2(-1.44269504088896340736f * x)); }

SLAM_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum of squares of the two bf16 values packed in w (what pack_bf16x2 produced: the values a bf16 store keeps)
SLAM_DEVICE float sq_bf16x2(uint32_t w) {
  const float a = __uint_as_float(w << 16), b = __uint_as_float(w & 0xffff0000u);
  return a * a + b * b;
}
// *slot = sum of s over the block's threads: lanes by the xor butterfly, waves in wave order - the same bits every run.
// `red`: WAVES floats of LDS nothing else touches at this point; EVERY thread of the block must call (one barrier inside).
template <int WAVES>
SLAM_DEVICE void block_sum_store(float s, float* red, float* slot) {
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += red[w];
    *slot = t;
  }
}
SLAM_DEVICE float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// D[i][j] += sum_k A[i][k] * B[k][j], 16x16x32 bf16.
//   a: lane l supplies A[i = l&15][k-block l>>4] (8 values)
//   b: lane l supplies B[k-block l>>4][j = l&15] (8 values)
//   d: lane l holds D[i = (l>>4)*4 + r][j = l&15], r = 0..3
SLAM_DEVICE f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// D[i][j] += sum_k A[i][k] * B[k][j], 32x32x16 bf16 (32 cycles per SIMD: the shape the part reaches its 2.5 PFLOP/s with;
// half the operand-register reads and half the issue slots per flop of the 16x16x32 form).
//   a: lane l supplies A[i = l&31][k = 8(l>>5) .. +7]
//   b: lane l supplies B[k = 8(l>>5) .. +7][j = l&31]
//   d: lane l holds D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31], r = 0..15
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
SLAM_DEVICE f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// LDS operand-tile layout shared by GEMM and attention: rows of 64 bf16 (128 B), eight
// 16-byte chunks per row, chunk index XOR-swizzled with key(row) = ((row>>1) ^ (row>>4)) & 7:
//  * fragment reads (ds_read_b128: 16 consecutive rows of a 16-aligned group, same chunk) see
//    (row>>1)&7 xor a group constant -> 16 distinct slots of the 256-byte bank row, conflict-free;
//  * direct / LDS-DMA staging writes one row's 8 chunks per 8-lane group -> conflict-free;
//  * transposed staging writes rows 8j + rr (j = 8 consecutive lanes) of one chunk per
//    ds_write_b128 8-lane group: key = (4(j&1) + (rr>>1)) ^ (j>>1) takes 8 distinct values, so
//    the (row>>4) term is what makes the wgrad/dgrad staging conflict-free as well.
SLAM_DEVICE int lds_swz_key(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }
SLAM_DEVICE int lds_tile_off(int row, int chunk) { return row * 128 + ((chunk ^ lds_swz_key(row)) << 4); }

// ---- LDS-DMA + transpose-read primitives (gfx950) -------------------------------------------------
// global_load_lds_dwordx4: 64 lanes x 16 B (each lane its own source address) land in LDS at
// M0 + lane*16. Issued from inline asm so hipcc's waitcnt insertion does not see it (it would drain
// it with vmcnt(0) before every ds_read); callers count completion with s_waitcnt vmcnt(N).
SLAM_DEVICE void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// same, source = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: the
// per-lane offsets are tile-invariant, so a K-loop only advances the scalar base (no VALU).
SLAM_DEVICE void glds16_sv(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
// same with the LDS destination as an m0 register constraint: the compiler materialises it with ONE s_mov / s_add
// instead of saving and restoring m0 around every load
SLAM_DEVICE void glds16_m0(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "{m0}"(lds_dst) : "memory");
}
SLAM_DEVICE void glds4_m0(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "{m0}"(lds_dst) : "memory");
}
// 4-byte variant: 64 lanes x 4 B -> LDS at M0 + lane*4
SLAM_DEVICE void glds4(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
SLAM_DEVICE void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
SLAM_DEVICE uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// ds_read_b64_tr_b16: within each 16-lane group, out[lane c][j] = in[lane 4j + (c>>2)][c&3] where
// in[lane i] are the 4 contiguous bf16 at lane i's address (measured: tools/probes/tr_probe.hip).
// With lane i pointing at X[k0 + (i>>2)][c0 + 4(i&3)..+3], lane c receives X[k0..k0+3][c0 + c]:
// four consecutive contraction rows of one column = half an MFMA operand.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
SLAM_DEVICE uint2 lds_tr_read(const char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
  return __builtin_bit_cast(uint2, v);
}

#define HIP_CHECK_RET(expr)                                     \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return (int)_e;                       \
  } while (0)
