// bf16 MFMA GEMM for gfx950: C[R,Cn] = A·Bᵀ with per-operand "contraction-major" staging.
//
// Replaces the torch Linear kernels behind Qwen2Attention.{q,k,v,o}_proj and
// Qwen2MLP.{gate,up,down}_proj (site-packages transformers/models/qwen2/modeling_qwen2.py:41-48,
// 189-192) and their autograd dgrad / wgrad; SURVEY.md §8a rows T3, T6, T7, T8.
//
//   forward : Y[M,N]  = X[M,K]  · W[N,K]ᵀ      (TA=0, TB=0)   both operands contraction-contiguous
//   dgrad   : dX[M,K] = dY[M,N] · W[N,K]       (TA=0, TB=1)   B stored [contraction][cols]
//             (the engine normally runs dgrad as the first form on a transposed weight image)
//   wgrad   : dW[N,K] = dYᵀ[N,M] · X[M,K]      (TA=1, TB=1)   A stored [contraction][rows]
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 fragments.
// Both operand tiles live in LDS as [128][64] bf16 with a 16-byte-chunk XOR swizzle (common.h), so
// the MFMA inner loop is identical for all forms; only global->LDS staging differs:
//   * LDS-DMA (global_load_lds_dwordx4 issued from inline asm so that hipcc does not drain it with
//     vmcnt(0) in front of every ds_read): NSTAGE-deep ring, counted s_waitcnt vmcnt(N), one
//     barrier per K-step; the swizzle is applied to the per-lane SOURCE address;
//   * register staging, direct or with an 8x8 in-register transpose (branch-free loads so the
//     next tile's loads stay in flight across the MFMA block).
// MFMA roles are swapped (a-operand = column tile, b-operand = row tile) so a lane ends up holding
// consecutive output columns of one row: four per fragment, and eight (16-byte accesses) in the
// default NT layout, where the column tile's rows are DMA'd in the perm64 order.
//
// Kernels in this file: gemm_kernel (128x128 tiles: NT with LDS-DMA, NN / TN / ragged shapes with register staging,
// every fused epilogue), gemm_nt_256_kernel (256x256 tiles, 8-phase schedule, for the wide-N launches),
// gemm_tn_bal_kernel + reduce_bal_kernel (wgrad, balanced K-splitting). The variants that lost their A/B in round 1
// (persistent blocks, 128x112 / 256x128 / 256x112 tiles, 128x128x32 tiles with deep rings, 8-wave 64x32 blocks,
// re-keyed swizzle) were removed in round 2; DESIGN.md section 4 keeps what each one measured.
#include <algorithm>
#include <type_traits>

#include <string.h>

#include "common.h"
#include "kernels.h"

namespace slam {
// ---- tuning state: one GemmTune per engine (kernels.h). The dispatch functions below read the CURRENT one, which an
// engine entry point installs for the duration of its call (GemmTuneScope in engine.hip); outside any engine call - the
// single-op entry points - the process default is current. Two engines in one process (DPO: policy + reference) no longer
// share or toggle each other's settings.
static GemmTune g_default_tune;
static thread_local GemmTune* t_cur_tune = nullptr;
GemmTune* gemm_default_tune() { return &g_default_tune; }
GemmTune* gemm_use_tune(GemmTune* t) { GemmTune* old = t_cur_tune; t_cur_tune = t; return old; }
GemmTune& T() { return t_cur_tune ? *t_cur_tune : g_default_tune; }
// The shipped library has no way to run a GEMM without its output stores: bit 1 of gemm_nt_store (the no-store probe of
// tools/probes/epilogue_cost.py / store_pmc.py) and the SLAM_PROBE_VMCNT override exist only in a -DSLAM_PROBES build
// (`python -m slamkit_amd.csrc.build --probes` -> lib/libslam_engine_probes.so, which only tools/probes/* load).
#ifdef SLAM_PROBES
#define SLAM_NOSTORE(nt) ((nt) & 2)
constexpr long NT_STORE_MAX = 3;
#else
#define SLAM_NOSTORE(nt) false
constexpr long NT_STORE_MAX = 1;
#endif
// 1 = set, 0 = unknown key, -1 = value outside the option's range (strict options only; the others clamp)
int gemm_tune_set(GemmTune* t, const char* key, long v) {
  auto clamp = [](long x, long lo, long hi) { return (int)(x < lo ? lo : x > hi ? hi : x); };
  if (!strcmp(key, "gemm_nt_store") && (v < 0 || v > NT_STORE_MAX)) return -1;
  struct { const char* k; int* f; long lo, hi; } tab[] = {
      {"gemm_glds", &t->glds, 0, 1}, {"gemm_tn_dma", &t->tn_dma, 0, 1}, {"gemm_group_rows", &t->group_rows, 1, 64},
      {"gemm_tn_splits", &t->tn_splits_override, 0, 64}, {"gemm_nt_store", &t->nt_store, 0, NT_STORE_MAX},
      {"gemm_256_persist", &t->g256_persist, 0, 1}, {"gemm_256", &t->g256, 0, 2}, {"gemm_nt224", &t->nt224, 0, 2},
      {"gemm_nt224_min_k", &t->nt224_min_k, 0, 1 << 30}, {"gemm_256_dswiglu", &t->g256_dswiglu, 0, 1},
      {"gemm_group_rows_256", &t->group_rows_256, 1, 64}, {"gemm_tn_balanced", &t->tn_balanced, 0, 1}, {"gemm_tn224", &t->tn224, 0, 2},
      {"gemm_tn224_min_m", &t->tn224_min_m, 0, 1 << 30}, {"gemm_tn224_max_split", &t->tn224_max_split, 1, 16},
      {"gemm_tn_bal_bg_max_split", &t->bal_bg_max_split, 1, 8}, {"gemm_tn224_bg_min_m", &t->tn224_bg_min_m, 0, 1 << 30},
      {"gemm_tn224_bg_max_split", &t->tn224_bg_max_split, 1, 16}, {"gemm_shared", &t->shared, 0, 1}, {"gemm_256_stagger", &t->g256_stagger, 0, 100000}, {"gemm_256_stagger_dswiglu", &t->g256_stagger_dswiglu, 0, 100000},
      {"gemm_256_cohorts", &t->g256_cohorts, 0, 32}, {"gemm_256_persist_cus", &t->g256_persist_cus, 0, 4096}, {"gemm_group_cols_256", &t->group_cols_256, 0, 4096}, {"gemm_mf32", &t->mf32, 0, 1}, {"gemm_256_w4", &t->g256_w4, 0, 1}, {"gemm_256_roles", &t->g256_roles, 0, 1}, {"gemm_256_batch_loads", &t->g256_batch_loads, 0, 1}};
  for (auto& e : tab)
    if (!strcmp(e.k, key)) { *e.f = clamp(v, e.lo, e.hi); return 1; }
  return 0;
}

}  // namespace slam
using slam::T;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;  // one operand tile: 128 rows x 128 B
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* resid;
  bf16_t* act;  // optional fused SwiGLU output [R][Cn/2] (4-wave kernel, 32-column gate/up blocks)
  bf16_t* gu;   // optional fused SwiGLU backward: the GEMM result is d(act) [R][Cn]; gu [R][2Cn] is rewritten in place with d(gate|up)
  int R, Cn, Kc;
  int lda, ldb, ldc;
  int kc_per_split;
  int tiles_r, tiles_c;
  // optional fused rotate-half RoPE on the first rope_heads 64-column heads (QKV projection epilogue):
  // fp32 cos/sin tables [R][32]
  const float* rope_cos;
  const float* rope_sin;
  int rope_heads;
  int group_rows;  // tile rasterisation group height (0/1 = plain row-major)
  int nt_store;    // streaming (non-temporal) output stores: large outputs must not evict the operand panels from L2
  const float* rope_cos_q;  // tables of the first rope_q_heads heads (queries: pre-scaled, kernels.h rope_table)
  const float* rope_sin_q;
  int rope_q_heads;
  int stagger_ticks;  // persistent 256 x 256 blocks with one tile fewer than the longest start this many 10-ns ticks late
  int cohorts;        // > 1: the slots of an XCD in this many contiguous cohorts, cohort c starts c * stagger_ticks late (every block)
  int group_cols;     // > 0: tile rasterisation groups are group_rows x group_cols tiles (0 = group_rows x all columns)
  int batch_epilogue_loads;  // persistent 256 x 256 SwiGLU backward: gate|up loads of a whole quadrant issued together (dswiglu_tile)
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
SLAM_DEVICE void st_out(bf16_t* ptr, const uint4& v, int nt) {
  u32x4_t w = {v.x, v.y, v.z, v.w};
  if (SLAM_NOSTORE(nt)) return;  // -DSLAM_PROBES builds only (tools/probes/epilogue_cost.py): the launch without its output stores
  if (nt) __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(ptr));
  else *reinterpret_cast<u32x4_t*>(ptr) = w;
}

SLAM_DEVICE uint32_t comp4(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
SLAM_DEVICE uint4 sel4(bool ok, const uint4& v) {
  return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
}

// ---- direct staging: operand stored [rows][contraction], 4 x 16 B per thread. Loads are
//      unconditional (out-of-range lanes read element 0 and are zeroed at store time) so no
//      branch / wait separates them. ---------------------------------------------------------
SLAM_DEVICE uint32_t load_direct(const bf16_t* G, int ld, int nrows, int row0, int k0, int kend, int tid,
                                 uint4* r) {
  uint32_t okm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + 256 * i;
    int row = q >> 3, c = q & 7;
    int gr = row0 + row, gk = k0 + c * 8;
    bool ok = (gr < nrows) & (gk < kend);
    size_t off = ok ? ((size_t)gr * ld + gk) : 0;
    r[i] = *reinterpret_cast<const uint4*>(G + off);
    okm |= (ok ? 1u : 0u) << i;
  }
  return okm;
}
SLAM_DEVICE void store_direct(char* tile, int tid, const uint4* r, uint32_t okm) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + 256 * i;
    int row = q >> 3, c = q & 7;
    *reinterpret_cast<uint4*>(tile + lds_tile_off(row, c)) = sel4((okm >> i) & 1, r[i]);
  }
}

// ---- transposed staging: operand stored [contraction][rows]; one 8(kc) x 8(rows) unit per thread,
//      128 units per tile (unit u: rows (u&15)*8.., kc (u>>4)*8..) -------------------------------
SLAM_DEVICE uint32_t load_transposed(const bf16_t* G, int ld, int nrows, int row0, int k0, int kend, int u,
                                     uint4* r) {
  int rb = u & 15, kb = u >> 4;
  int gr = row0 + rb * 8;
  uint32_t okm = 0;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    int gk = k0 + kb * 8 + kk;
    bool ok = (gk < kend) & (gr < nrows);
    size_t off = ok ? ((size_t)gk * ld + gr) : 0;
    r[kk] = *reinterpret_cast<const uint4*>(G + off);
    okm |= (ok ? 1u : 0u) << kk;
  }
  return okm;
}
SLAM_DEVICE void store_transposed(char* tile, int u, const uint4* rin, uint32_t okm) {
  int rb = u & 15, kb = u >> 4;
  uint4 r[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) r[kk] = sel4((okm >> kk) & 1, rin[kk]);
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    uint32_t w[4];
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) {
      uint32_t a = comp4(r[2 * wi], rr >> 1), b = comp4(r[2 * wi + 1], rr >> 1);
      w[wi] = (rr & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    int row = rb * 8 + rr;
    *reinterpret_cast<uint4*>(tile + lds_tile_off(row, kb)) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---- LDS-DMA staging (direct operands only). One global_load_lds_dwordx4 moves 64 lanes x 16 B to
//      LDS at M0 + lane*16; issued from asm so the compiler's waitcnt insertion does not see it
//      (we count it ourselves with s_waitcnt vmcnt(N)). Rows past the end are clamped: their
//      products only reach output rows that are never stored. ---------------------------------
// Per-lane byte offsets (tile-invariant) of the 4 chunks a lane moves, RELATIVE to the tile's first row (a 32-bit
// offset from the matrix origin wraps once rows x ld x 2 B passes 4 GB: dlogits [16384][152320] is 4.99 GB); the K-loop
// only advances the wave-uniform 64-bit base pointer G + row0 * ld + k0.
template <int THREADS, int ROWS>
SLAM_DEVICE void glds_offsets(int ld, int nrows, int row0, int tid, uint32_t* voff) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / THREADS; ++i) {
    int P = i * THREADS + tid;
    int row = P >> 3, cs = P & 7;
    int c = cs ^ lds_swz_key(row);
    int gr = row0 + row;
    gr = (gr < nrows ? gr : nrows - 1) - row0;  // relative to the tile's first row: the 32-bit offset must not span the matrix
    voff[i] = (uint32_t)(((size_t)gr * ld + c * 8) * sizeof(bf16_t));
  }
}
// Column-tile fragment assignment for 16-byte epilogue accesses: LDS row v of a 64-row block of the column tile holds
// tile row perm64(v) (the permutation is applied to the DMA SOURCE rows; LDS layout, swizzle and fragment reads are those
// of the natural order), so that the fragments fn = 2q, 2q+1 of a lane (MFMA rows i = 4g..4g+3 of each) are the 8
// CONSECUTIVE output columns 32q + 8g .. +7: half the store / bias / residual instructions of the 4-column layout.
SLAM_DEVICE int perm64(int v) {
  const int fn = v >> 4, i = v & 15;
  return ((fn >> 1) << 5) | ((i >> 2) << 3) | ((fn & 1) << 2) | (i & 3);
}
template <int THREADS, int ROWS>
SLAM_DEVICE void glds_offsets_perm(int ld, int nrows, int row0, int tid, uint32_t* voff) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / THREADS; ++i) {
    int P = i * THREADS + tid;
    int row = P >> 3, cs = P & 7;
    int c = cs ^ lds_swz_key(row);
    int gr = row0 + (row & ~63) + perm64(row & 63);
    gr = (gr < nrows ? gr : nrows - 1) - row0;
    voff[i] = (uint32_t)(((size_t)gr * ld + c * 8) * sizeof(bf16_t));
  }
}
// The same idea for the 32x32x16 MFMA (round 5): a 32-row block of the column tile is the MFMA's row index i, and register
// r = 4q + e of lane half h = lane >> 5 holds i = e + 4h + 8q. LDS row v of the block holds tile column perm32(v) =
// 16h + 4q + e, so a lane ends up with the SIXTEEN consecutive output columns 16h .. 16h + 15 of one output row per block.
SLAM_DEVICE int perm32(int v) { return (((v >> 2) & 1) << 4) | ((v >> 3) << 2) | (v & 3); }
template <int THREADS, int ROWS>
SLAM_DEVICE void glds_offsets_perm32(int ld, int nrows, int row0, int tid, uint32_t* voff) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / THREADS; ++i) {
    int P = i * THREADS + tid;
    int row = P >> 3, cs = P & 7;
    int c = cs ^ lds_swz_key(row);
    int gr = row0 + (row & ~31) + perm32(row & 31);
    gr = (gr < nrows ? gr : nrows - 1) - row0;
    voff[i] = (uint32_t)(((size_t)gr * ld + c * 8) * sizeof(bf16_t));
  }
}
template <int THREADS, int ROWS>
SLAM_DEVICE void glds_tile(const bf16_t* Gk /* = G + k0, wave-uniform */, const uint32_t* voff, int wave,
                           uint32_t tile_lds) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / THREADS; ++i)
    glds16_sv(Gk, voff[i], __builtin_amdgcn_readfirstlane(tile_lds + (uint32_t)(i * THREADS + wave * 64) * 16u));
}

// ---- transposed operands through LDS-DMA + hardware transpose reads (wgrad). The operand is
//      stored [contraction][rows] in global memory and lands in LDS the same way ([64 kc][128 rows],
//      256-B rows); ds_read_b64_tr_b16 then hands lane c of each 16-lane group four consecutive kc
//      values of row c (measured semantics, tools/probes/tr_probe.hip: out[lane c][j] =
//      in[lane 4j + (c>>2)][c&3], every lane supplying the address of 4 contiguous bf16).
//      32-byte row-blocks are XOR-swizzled with key(kc) = (kc&3) | ((kc>>3)&1)<<2 so the 8 kc rows
//      a 32-lane group touches fall into 8 distinct 32-B windows of the 256-B bank row. ----------
SLAM_DEVICE int tr_key(int kc) { return (kc & 3) | (((kc >> 3) & 1) << 2); }

template <int THREADS, int ROWS>
SLAM_DEVICE void glds_offsets_tr(int ld, int row0, int tid, uint32_t* voff) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / THREADS; ++i) {
    int P = i * THREADS + tid;
    int kc = P / (ROWS / 8), cs = P % (ROWS / 8);
    int c = cs ^ (tr_key(kc) << 1);
    voff[i] = (uint32_t)(((size_t)kc * ld + row0 + c * 8) * sizeof(bf16_t));
  }
}

// Epilogue of the 8-column layout: lane (l15, g) of wave (wm, wn) holds C[m][cw + 32q .. +7] for
// q = 0, 1 in fragments (2q, 2q+1): 16-byte accesses throughout. Fused: bias, residual, RoPE, SwiGLU fwd / bwd.
// LEAN: plain / SwiGLU forward / SwiGLU backward only (no bias, residual or RoPE operands compiled in) - the persistent
// 256 x 256 kernel must not carry a conditionally consumed load around its tile loop: the compiler would wait for it
// (vmcnt) inside the next tile's first K-tile, and that wait also covers the epilogue's stores.
template <bool LEAN = false>
SLAM_DEVICE void epilogue8(const GemmArgs& p_, const f32x4_t (&acc)[4][4], int row0, int col0, int wm, int wn, int l15, int g) {
  struct Fields {  // what the fused paths read; LEAN pins the optional operands to null at compile time
    void* C; bf16_t* act; bf16_t* gu; int R, Cn, ldc, nt_store;
    const bf16_t* bias; const bf16_t* resid; const float* rope_cos; const float* rope_sin; int rope_heads;
    const float* rope_cos_q; const float* rope_sin_q; int rope_q_heads;
  };
  const Fields p = {p_.C, p_.act, p_.gu, p_.R, p_.Cn, p_.ldc, p_.nt_store,
                  LEAN ? nullptr : p_.bias, LEAN ? nullptr : p_.resid, LEAN ? nullptr : p_.rope_cos,
                  LEAN ? nullptr : p_.rope_sin, LEAN ? 0 : p_.rope_heads,
                  LEAN ? nullptr : p_.rope_cos_q, LEAN ? nullptr : p_.rope_sin_q, LEAN ? 0 : p_.rope_q_heads};
  // lane holds C[m][cq(q) .. +7] for q = 0, 1 in fragments (2q, 2q+1): 16-byte accesses throughout
  const int cw = col0 + wn * 64 + g * 8;  // + 32 q
  uint4 bb4[2];
  if (p.bias) {
#pragma unroll
    for (int q = 0; q < 2; ++q) bb4[q] = *reinterpret_cast<const uint4*>(p.bias + cw + 32 * q);
  }
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int m = row0 + wm * 64 + fm * 16 + l15;
    const bool mok = m < p.R;
    const size_t rowoff = (size_t)(mok ? m : 0) * p.ldc;
    if (p.gu) {
      // fused SwiGLU backward: acc = d(act)[m][c..c+7]; gate at gu[m][(c/32)*64 + c%32], up 32 columns later
      bf16_t* grow = p.gu + (size_t)(mok ? m : 0) * (2 * p.Cn);
      uint4 gg[2], uu[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = cw + 32 * q;
        const bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
        gg[q] = *reinterpret_cast<const uint4*>(gp);
        uu[q] = *reinterpret_cast<const uint4*>(gp + 32);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = cw + 32 * q;
        float gv[8], uv[8], dg[8], du[8];
        unpack_bf16x8(gg[q], gv);
        unpack_bf16x8(uu[q], uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = acc[fm][2 * q + (e >> 2)][e & 3];
          const float sg = fast_sigmoid(gv[e]);
          du[e] = d * gv[e] * sg;
          dg[e] = d * uv[e] * sg * (1.f + gv[e] * (1.f - sg));
        }
        if (mok && !SLAM_NOSTORE(p.nt_store)) {
          bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
          *reinterpret_cast<uint4*>(gp) = pack_bf16x8(dg);
          *reinterpret_cast<uint4*>(gp + 32) = pack_bf16x8(du);
        }
      }
      continue;
    }
    uint4 rr4[2];
    if (p.resid) {
#pragma unroll
      for (int q = 0; q < 2; ++q) rr4[q] = *reinterpret_cast<const uint4*>(p.resid + rowoff + cw + 32 * q);
    }
    float v[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[q][e] = acc[fm][2 * q + (e >> 2)][e & 3];
      if (p.bias) {
        float b[8];
        unpack_bf16x8(bb4[q], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][e] += b[e];
      }
      if (p.resid) {
        float r[8];
        unpack_bf16x8(rr4[q], r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][e] += r[e];
      }
    }
    // fused RoPE: the wave's 64 columns are one head; q = 0 / 1 hold d and d + 32 of the same lane
    if (p.rope_cos && mok && ((col0 + wn * 64) >> 6) < p.rope_heads) {
      const bool qh = ((col0 + wn * 64) >> 6) < p.rope_q_heads;  // wave-uniform: the wave's 64 columns are one head
      const float4* cp = reinterpret_cast<const float4*>((qh ? p.rope_cos_q : p.rope_cos) + (size_t)m * 32 + g * 8);
      const float4* sp = reinterpret_cast<const float4*>((qh ? p.rope_sin_q : p.rope_sin) + (size_t)m * 32 + g * 8);
      float cc[8], ss[8];
      *reinterpret_cast<float4*>(cc) = cp[0]; *reinterpret_cast<float4*>(cc + 4) = cp[1];
      *reinterpret_cast<float4*>(ss) = sp[0]; *reinterpret_cast<float4*>(ss + 4) = sp[1];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x1 = v[0][e], x2 = v[1][e];
        v[0][e] = x1 * cc[e] - x2 * ss[e];
        v[1][e] = x2 * cc[e] + x1 * ss[e];
      }
    }
    if (mok) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        st_out(reinterpret_cast<bf16_t*>(p.C) + rowoff + cw + 32 * q, pack_bf16x8(v[q]), p.nt_store);
      // fused SwiGLU: this wave's 64 columns are [32 gate | 32 up] = (q 0 | q 1) of the same lane
      if (p.act) {
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gt = acc[fm][e >> 2][e & 3], up = acc[fm][2 + (e >> 2)][e & 3];
          a8[e] = gt * fast_sigmoid(gt) * up;
        }
        st_out(p.act + (size_t)m * (p.Cn / 2) + (col0 + wn * 64) / 2 + g * 8, pack_bf16x8(a8), p.nt_store);
      }
    }
  }
}

// ---- fused SwiGLU backward of a whole 128 x 64 wave tile with its gate|up loads batched (round 6) ---------------------------------
// epilogue8's SwiGLU-backward path walks one 16-row fragment at a time: 4 loads -> wait -> arithmetic -> 4 stores, eight times per
// tile and wave. The vector-memory path of a CU is in order, so every one of those load groups queues behind the stores of the
// fragment before it: eight exposed round trips per tile, each as long as the store queue in front of it (ISA of round 5:
// `global_load_dwordx4` x4, `s_waitcnt vmcnt(3)`, ..., `global_store_dwordx4` x4, repeat). Here the 16 loads of a 64-row
// quadrant are issued together, and the second quadrant's loads BEFORE the first quadrant's stores (they go into the registers
// the first quadrant's accumulators leave behind): two load round trips per tile, neither behind a store. Same arithmetic, same
// bits (tests: test_gemm_nt_256_persistent_blocks compares with the one-block-per-tile kernel, which keeps epilogue8).
SLAM_DEVICE void dswiglu_tile(const GemmArgs& p, const f32x4_t (&acc)[2][4][4], int row0, int col0, int wn, int l15, int g) {
  const int cw = col0 + wn * 64 + g * 8;
  uint4 gg[2][4][2], uu[2][4][2];  // [quadrant][fm][q]; overwritten in place with the packed d gate / d up
  auto gptr = [&](int mq, int fm, int q) -> bf16_t* {
    const int c = cw + 32 * q;
    return p.gu + (size_t)(row0 + mq * 64 + fm * 16 + l15) * (2 * p.Cn) + (c >> 5) * 64 + (c & 31);
  };
  auto load = [&](int mq) {
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bf16_t* gp = gptr(mq, fm, q);
        gg[mq][fm][q] = *reinterpret_cast<const uint4*>(gp);
        uu[mq][fm][q] = *reinterpret_cast<const uint4*>(gp + 32);
      }
  };
  auto compute = [&](int mq) {
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float gv[8], uv[8], dg[8], du[8];
        unpack_bf16x8(gg[mq][fm][q], gv);
        unpack_bf16x8(uu[mq][fm][q], uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = acc[mq][fm][2 * q + (e >> 2)][e & 3];
          const float sg = fast_sigmoid(gv[e]);
          du[e] = d * gv[e] * sg;
          dg[e] = d * uv[e] * sg * (1.f + gv[e] * (1.f - sg));
        }
        gg[mq][fm][q] = pack_bf16x8(dg);
        uu[mq][fm][q] = pack_bf16x8(du);
      }
  };
  auto store = [&](int mq) {
    if (SLAM_NOSTORE(p.nt_store)) return;
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16_t* gp = gptr(mq, fm, q);
        *reinterpret_cast<uint4*>(gp) = gg[mq][fm][q];
        *reinterpret_cast<uint4*>(gp + 32) = uu[mq][fm][q];
      }
  };
  load(0);
  __builtin_amdgcn_sched_barrier(0);
  compute(0);
  __builtin_amdgcn_sched_barrier(0);
  load(1);  // ahead of quadrant 0's stores in the CU's in-order memory path
  __builtin_amdgcn_sched_barrier(0);
  store(0);
  __builtin_amdgcn_sched_barrier(0);
  compute(1);
  store(1);
}

// ---- role-split epilogue of the persistent 256 x 256 kernel (round 6, "gemm_256_roles") -------------------------------------------
// One 64-row quadrant of a wave's 128 x 64 output as a list of 16-byte RECORDS - exactly what epilogue8<LEAN> stores, in its
// order and with its arithmetic: plain [fm][q] {C}; SwiGLU forward [fm][q] {C} then [fm] {act}; SwiGLU backward [fm][q]
// {d gate, d up}. quad_emit computes them (TO_LDS: into the lane's slots of a staging region instead of memory); quad_drain
// reads staged records and stores them where quad_emit<false> of the producing wave would have - the two loop nests are the
// same by construction.
SLAM_DEVICE uint4 lds_rec_read(const char* p) { return *reinterpret_cast<const uint4*>(p); }
template <bool TO_LDS>
SLAM_DEVICE void quad_emit(const GemmArgs& p, const f32x4_t (&acc)[4][4], int row0, int col0, int wn, int l15, int g, char* stg) {
  const int cw = col0 + wn * 64 + g * 8;
  auto out = [&](int k, bf16_t* ptr, const uint4& v) {
    if (TO_LDS) *reinterpret_cast<uint4*>(stg + k * 1024) = v;
    else st_out(ptr, v, p.nt_store);
  };
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int m = row0 + fm * 16 + l15;
    const size_t rowoff = (size_t)m * p.ldc;
    if (p.gu) {
      bf16_t* grow = p.gu + (size_t)m * (2 * p.Cn);
      uint4 gg[2], uu[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = cw + 32 * q;
        const bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
        gg[q] = *reinterpret_cast<const uint4*>(gp);
        uu[q] = *reinterpret_cast<const uint4*>(gp + 32);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = cw + 32 * q;
        float gv[8], uv[8], dg[8], du[8];
        unpack_bf16x8(gg[q], gv);
        unpack_bf16x8(uu[q], uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = acc[fm][2 * q + (e >> 2)][e & 3];
          const float sg = fast_sigmoid(gv[e]);
          du[e] = d * gv[e] * sg;
          dg[e] = d * uv[e] * sg * (1.f + gv[e] * (1.f - sg));
        }
        bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
        out((fm * 2 + q) * 2, gp, pack_bf16x8(dg));
        out((fm * 2 + q) * 2 + 1, gp + 32, pack_bf16x8(du));
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[fm][2 * q + (e >> 2)][e & 3];
      out(fm * 2 + q, reinterpret_cast<bf16_t*>(p.C) + rowoff + cw + 32 * q, pack_bf16x8(v));
    }
    if (p.act) {
      float a8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gt = acc[fm][e >> 2][e & 3], up = acc[fm][2 + (e >> 2)][e & 3];
        a8[e] = gt * fast_sigmoid(gt) * up;
      }
      out(8 + fm, p.act + (size_t)m * (p.Cn / 2) + (col0 + wn * 64) / 2 + g * 8, pack_bf16x8(a8));
    }
  }
}
SLAM_DEVICE void quad_drain(const GemmArgs& p, int row0, int col0, int wn, int l15, int g, const char* stg) {
  const int cw = col0 + wn * 64 + g * 8;
  if (p.gu) {
    uint4 r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = lds_rec_read(stg + k * 1024);
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
      bf16_t* grow = p.gu + (size_t)(row0 + fm * 16 + l15) * (2 * p.Cn);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = cw + 32 * q;
        bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
        st_out(gp, r[(fm * 2 + q) * 2], p.nt_store);
        st_out(gp + 32, r[(fm * 2 + q) * 2 + 1], p.nt_store);
      }
    }
    return;
  }
  uint4 r[12];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = lds_rec_read(stg + k * 1024);
  if (p.act) {
#pragma unroll
    for (int k = 8; k < 12; ++k) r[k] = lds_rec_read(stg + k * 1024);
  }
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int m = row0 + fm * 16 + l15;
#pragma unroll
    for (int q = 0; q < 2; ++q) st_out(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + cw + 32 * q, r[fm * 2 + q], p.nt_store);
    if (p.act) st_out(p.act + (size_t)m * (p.Cn / 2) + (col0 + wn * 64) / 2 + g * 8, r[8 + fm], p.nt_store);
  }
}

// Epilogue of the 32x32x16 kernels: a 64 x 64 piece of the output held as acc[rb][nb] (32-row block rb, 32-column block nb
// of the perm32 layout): lane (l31, h) holds C[row0 + 32 rb + l31][cbase + 32 nb + 16 h .. +15] in the 16 registers of a
// block - two 16-byte accesses per block and operand. cbase is a multiple of 64: the 64 columns are one attention head
// (RoPE: nb = 0 / 1 hold d and d + 32 of the same lane) or one [32 gate | 32 up] block (SwiGLU forward: nb = 0 / 1 are
// gate and up of the same 16 activation columns). Same fused paths and the same arithmetic per element as epilogue8.
template <bool LEAN = false>
SLAM_DEVICE void epilogue32(const GemmArgs& p_, const f32x16_t (&acc)[2][2], int row0, int cbase, int l31, int h) {
  struct Fields {
    void* C; bf16_t* act; bf16_t* gu; int R, Cn, ldc, nt_store;
    const bf16_t* bias; const bf16_t* resid; const float* rope_cos; const float* rope_sin; int rope_heads;
    const float* rope_cos_q; const float* rope_sin_q; int rope_q_heads;
  };
  const Fields p = {p_.C, p_.act, p_.gu, p_.R, p_.Cn, p_.ldc, p_.nt_store,
                  LEAN ? nullptr : p_.bias, LEAN ? nullptr : p_.resid, LEAN ? nullptr : p_.rope_cos,
                  LEAN ? nullptr : p_.rope_sin, LEAN ? 0 : p_.rope_heads,
                  LEAN ? nullptr : p_.rope_cos_q, LEAN ? nullptr : p_.rope_sin_q, LEAN ? 0 : p_.rope_q_heads};
  const int cl = cbase + 16 * h;  // + 32 nb
  uint4 bb4[2][2];
  if (p.bias) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int u = 0; u < 2; ++u) bb4[nb][u] = *reinterpret_cast<const uint4*>(p.bias + cl + 32 * nb + 8 * u);
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int m = row0 + rb * 32 + l31;
    const bool mok = m < p.R;
    const size_t rowoff = (size_t)(mok ? m : 0) * p.ldc;
    if (p.gu) {
      // fused SwiGLU backward: acc = d(act)[m][c .. c+15]; gate at gu[m][(c/32)*64 + c%32], up 32 columns later
      bf16_t* grow = p.gu + (size_t)(mok ? m : 0) * (2 * p.Cn);
      uint4 gg[2][2], uu[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int c = cl + 32 * nb;
        const bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          gg[nb][u] = *reinterpret_cast<const uint4*>(gp + 8 * u);
          uu[nb][u] = *reinterpret_cast<const uint4*>(gp + 32 + 8 * u);
        }
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int c = cl + 32 * nb;
        bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float gv[8], uv[8], dg[8], du[8];
          unpack_bf16x8(gg[nb][u], gv);
          unpack_bf16x8(uu[nb][u], uv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = acc[rb][nb][8 * u + e];
            const float sg = fast_sigmoid(gv[e]);
            du[e] = d * gv[e] * sg;
            dg[e] = d * uv[e] * sg * (1.f + gv[e] * (1.f - sg));
          }
          if (mok && !SLAM_NOSTORE(p.nt_store)) {
            *reinterpret_cast<uint4*>(gp + 8 * u) = pack_bf16x8(dg);
            *reinterpret_cast<uint4*>(gp + 32 + 8 * u) = pack_bf16x8(du);
          }
        }
      }
      continue;
    }
    uint4 rr4[2][2];
    if (p.resid) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u) rr4[nb][u] = *reinterpret_cast<const uint4*>(p.resid + rowoff + cl + 32 * nb + 8 * u);
    }
    float v[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[nb][e] = acc[rb][nb][e];
      if (p.bias) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float b[8];
          unpack_bf16x8(bb4[nb][u], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[nb][8 * u + e] += b[e];
        }
      }
      if (p.resid) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float r[8];
          unpack_bf16x8(rr4[nb][u], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[nb][8 * u + e] += r[e];
        }
      }
    }
    if (p.rope_cos && mok && (cbase >> 6) < p.rope_heads) {
      const bool qh = (cbase >> 6) < p.rope_q_heads;  // wave-uniform: the 64 columns are one head
      const float4* cp = reinterpret_cast<const float4*>((qh ? p.rope_cos_q : p.rope_cos) + (size_t)m * 32 + h * 16);
      const float4* sp = reinterpret_cast<const float4*>((qh ? p.rope_sin_q : p.rope_sin) + (size_t)m * 32 + h * 16);
      float cc[16], ss[16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        *reinterpret_cast<float4*>(cc + 4 * u) = cp[u];
        *reinterpret_cast<float4*>(ss + 4 * u) = sp[u];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float x1 = v[0][e], x2 = v[1][e];
        v[0][e] = x1 * cc[e] - x2 * ss[e];
        v[1][e] = x2 * cc[e] + x1 * ss[e];
      }
    }
    if (mok) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          st_out(reinterpret_cast<bf16_t*>(p.C) + rowoff + cl + 32 * nb + 8 * u, pack_bf16x8(v[nb] + 8 * u), p.nt_store);
      if (p.act) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float a8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gt = acc[rb][0][8 * u + e], up = acc[rb][1][8 * u + e];
            a8[e] = gt * fast_sigmoid(gt) * up;
          }
          st_out(p.act + (size_t)m * (p.Cn / 2) + cbase / 2 + 16 * h + 8 * u, pack_bf16x8(a8), p.nt_store);
        }
      }
    }
  }
}

// 4 waves as 2x2 of 64x64, two blocks per CU. GLDS: two-stage LDS-DMA ring (operands whose rows / contraction are whole
// tiles); otherwise register staging with bounds handling (ragged shapes, NN dgrad without a transposed image).
// PERM: 8-column epilogue layout (NT DMA form, bf16 output).
// MF32 (with PERM): the main loop on v_mfma_f32_32x32x16_bf16 - a wave's 64 x 64 are 2 x 2 blocks of 32 x 32, the column
// tile's rows in the perm32 order, epilogue32.
template <bool TA, bool TB, bool F32OUT, bool GLDS, bool PERM = false, bool MF32 = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int THREADS = 256, NSTAGE = 2, BMT = BM;
  constexpr int WNN = 2;  // wave grid 2 x 2
  constexpr int NF = 4;   // 16-column fragments per wave
  constexpr int A_BYTES = TILE_BYTES, STAGE = STAGE_BYTES;
  static_assert(!PERM || (GLDS && !(TA && TB) && !F32OUT), "8-column layout: NT DMA kernel, bf16 out");
  static_assert(!MF32 || PERM, "32x32x16 main loop: NT DMA kernel only");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNN, wn = wave % WNN;
  const int l15 = lane & 15, g = lane >> 4;
  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles
  // (bijective for any tile count).
  const int nblk = p.tiles_r * p.tiles_c;
  int nid;
  {
    int id = blockIdx.x, xcd = id & 7, idx = id >> 3;
    int q = nblk >> 3, r = nblk & 7;
    nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // grouped rasterisation inside an XCD's run: GR row-tiles x all column tiles per group, column-major
  // inside the group, so the tiles in flight on one XCD share GR row panels and a few column panels
  // (the weight matrix of the wide projections does not fit the 4 MB L2: measured 590 MB of L2-miss
  // reads per gate|up launch with row-major order, 32 MB algorithmic)
  int tr_, tc_;
  {
    const int GR = p.group_rows > 0 ? p.group_rows : 1;
    const int per_group = GR * p.tiles_c;
    const int grp = nid / per_group, in = nid - grp * per_group;
    const int rows_here = min(GR, p.tiles_r - grp * GR);  // last group may be short
    tc_ = in / rows_here;
    tr_ = grp * GR + in - tc_ * rows_here;
  }
  const int row0 = tr_ * BMT;
  const int col0 = tc_ * BN;
  const int kbeg = blockIdx.z * p.kc_per_split;
  const int kend = min(p.Kc, kbeg + p.kc_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x4_t acc[4][NF];     // 16x16x32 form (dead in the MF32 instantiation)
  f32x16_t acc32[2][2];   // 32x32x16 form: [32-row block][32-column block] (dead otherwise)
  if constexpr (MF32) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  // 32x32x16 fragments: lane (l31, hh) reads row (w*64 + b*32 + l31), K-step s: chunk (2s + hh) ^ key(row) with
  // key = ((l31>>1) ^ (l31>>4) ^ (w*4 + b*2)) & 7 (common.h: 16 distinct 16-byte slots per ds_read_b128 lane group)
  const int l31 = lane & 31, hh = lane >> 5;
  const int k32 = ((l31 >> 1) ^ (l31 >> 4)) & 7;
  auto compute32 = [&](int s_) {
    const char* At = smem + s_ * STAGE;
    const char* Bt = At + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 af[2], bf[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        af[b] = *reinterpret_cast<const uint4*>(Bt + (wn * 64 + b * 32 + l31) * 128 + ((((2 * ks + hh) ^ k32 ^ (wn * 4 + b * 2)) & 7) << 4));
        bf[b] = *reinterpret_cast<const uint4*>(At + (wm * 64 + b * 32 + l31) * 128 + ((((2 * ks + hh) ^ k32 ^ (wm * 4 + b * 2)) & 7) << 4));
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc32[rb][nb] = mfma32(af[nb], bf[rb], acc32[rb][nb]);
    }
  };

  // swizzle key of row (w*64 + f*16 + l15) = ((l15>>1) ^ (w*4 + f)) & 7 = s0 ^ f
  const int s0a = ((l15 >> 1) ^ (wn * NF)) & 7;
  const int s0b = ((l15 >> 1) ^ (wm * 4)) & 7;
  const int a_base = (wn * NF * 16 + l15) * 128;  // a-operand = column (B) tile
  const int b_base = (wm * 64 + l15) * 128;  // b-operand = row (A) tile

  auto compute = [&](int s) {
    const char* At = smem + s * STAGE;
    const char* Bt = At + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ca = ((g + 4 * kk) ^ s0a) << 4, cb = ((g + 4 * kk) ^ s0b) << 4;
      uint4 af[NF], bf[4];
#pragma unroll
      for (int f = 0; f < NF; ++f)
        af[f] = *reinterpret_cast<const uint4*>(Bt + a_base + f * 16 * 128 + (ca ^ (f << 4)));
#pragma unroll
      for (int f = 0; f < 4; ++f)
        bf[f] = *reinterpret_cast<const uint4*>(At + b_base + f * 16 * 128 + (cb ^ (f << 4)));
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < NF; ++fn) acc[fm][fn] = mfma16(af[fn], bf[fm], acc[fm][fn]);
    }
  };

  // transposed-operand fragments: lane (l15, g) of fragment P (16 rows) at k-step kk reads
  // kc = kk*32 + g*8 + h*4 + (l15>>2), 8 bytes at row-block (P ^ key) and sub-offset (l15&3)*8
  const int trk = (l15 >> 2) | ((g & 1) << 2);
  constexpr int PA = BMT * 2;  // bytes per kc row of the transposed A tile ([64 kc][BMT rows])
  const int tr_lane = (g * 8 + (l15 >> 2)) * 256 + (l15 & 3) * 8;
  const int tr_lane_a = (g * 8 + (l15 >> 2)) * PA + (l15 & 3) * 8;
  auto compute_tr = [&](int s) {
    const char* At = smem + s * STAGE;
    const char* Bt = At + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[NF], bf[4];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const char* pa = Bt + tr_lane + kk * 32 * 256 + (((wn * NF + f) ^ trk) << 5);
        uint2 a0 = lds_tr_read(pa), a1 = lds_tr_read(pa + 4 * 256);
        af[f] = make_uint4(a0.x, a0.y, a1.x, a1.y);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const char* pb = At + tr_lane_a + kk * 32 * PA + (((wm * 4 + f) ^ trk) << 5);
        uint2 b0 = lds_tr_read(pb), b1 = lds_tr_read(pb + 4 * PA);
        bf[f] = make_uint4(b0.x, b0.y, b1.x, b1.y);
      }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < NF; ++fn) acc[fm][fn] = mfma16(af[fn], bf[fm], acc[fm][fn]);
    }
  };

  if constexpr (GLDS) {
    constexpr int D = NSTAGE - 1;  // tiles in flight ahead of the one being computed
    constexpr bool TR = TA && TB;  // both operands stored [contraction][rows]: DMA + transpose reads
    const uint32_t lds0 = lds_addr(smem);
    uint32_t voa[4], vob[4];
    if constexpr (TR) {
      glds_offsets_tr<THREADS, BMT>(p.lda, row0, tid, voa);
      glds_offsets_tr<THREADS, BN>(p.ldb, col0, tid, vob);
    } else {
      glds_offsets<THREADS, BMT>(p.lda, p.R, row0, tid, voa);
      if constexpr (MF32) glds_offsets_perm32<THREADS, BN>(p.ldb, p.Cn, col0, tid, vob);
      else if constexpr (PERM) glds_offsets_perm<THREADS, BN>(p.ldb, p.Cn, col0, tid, vob);
      else glds_offsets<THREADS, BN>(p.ldb, p.Cn, col0, tid, vob);
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int t) {
      const int k0 = kbeg + t * BK;
      const uint32_t st = lds0 + (uint32_t)((t % NSTAGE) * STAGE);
      // wave-uniform bases: direct operands advance along the contiguous contraction dim, transposed
      // operands by whole rows
      const bf16_t* ga = TR ? p.A + (size_t)k0 * p.lda : p.A + (size_t)row0 * p.lda + k0;
      const bf16_t* gb = TR ? p.B + (size_t)k0 * p.ldb : p.B + (size_t)col0 * p.ldb + k0;
      glds_tile<THREADS, BMT>(ga, voa, wv, st);
      glds_tile<THREADS, BN>(gb, vob, wv, st + A_BYTES);
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nk) issue(s);
    for (int t = 0; t < nk; ++t) {
      wait_vmcnt<0>();   // tile t (the only one in flight) has landed
      __syncthreads();  // everyone's tile-t DMAs landed; everyone is done reading stage (t-1)%NSTAGE
      if (t + D < nk) issue(t + D);
      if constexpr (TR) compute_tr(t % NSTAGE);
      else if constexpr (MF32) compute32(t % NSTAGE);
      else compute(t % NSTAGE);
    }
  } else {
    uint4 sa[8], sb[8];
    uint32_t ma = 0, mb = 0;
    auto stage_load = [&](int t) {
      const int k0 = kbeg + t * BK;
      if constexpr (TA && TB) {
        if (tid < 128) ma = load_transposed(p.A, p.lda, p.R, row0, k0, kend, tid, sa);
        else ma = load_transposed(p.B, p.ldb, p.Cn, col0, k0, kend, tid - 128, sa);
      } else {
        if constexpr (TA) { if (tid < 128) ma = load_transposed(p.A, p.lda, p.R, row0, k0, kend, tid, sa); }
        else ma = load_direct(p.A, p.lda, p.R, row0, k0, kend, tid, sa);
        if constexpr (TB) { if (tid < 128) mb = load_transposed(p.B, p.ldb, p.Cn, col0, k0, kend, tid, sb); }
        else mb = load_direct(p.B, p.ldb, p.Cn, col0, k0, kend, tid, sb);
      }
    };
    auto stage_store = [&](int s) {
      char* At = smem + s * STAGE_BYTES;
      char* Bt = At + TILE_BYTES;
      if constexpr (TA && TB) {
        if (tid < 128) store_transposed(At, tid, sa, ma);
        else store_transposed(Bt, tid - 128, sa, ma);
      } else {
        if constexpr (TA) { if (tid < 128) store_transposed(At, tid, sa, ma); }
        else store_direct(At, tid, sa, ma);
        if constexpr (TB) { if (tid < 128) store_transposed(Bt, tid, sb, mb); }
        else store_direct(Bt, tid, sb, mb);
      }
    };
    if (nk > 0) {
      stage_load(0);
      stage_store(0);
      __syncthreads();
      for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) stage_load(t + 1);
        compute(t & 1);
        if (t + 1 < nk) stage_store((t + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: lane holds C[m][n..n+3] for each (fm, fn); bias / residual loads are batched
  //      per row so they are all in flight together -------------------------------------------
  if constexpr (MF32) {
    epilogue32(p, acc32, row0 + wm * 64, col0 + wn * 64, l31, hh);
    return;
  } else if constexpr (PERM) {
    epilogue8(p, acc, row0, col0, wm, wn, l15, g);
    return;
  }
  uint2 bb[4];
  if (!F32OUT && p.bias) {
#pragma unroll
    for (int fn = 0; fn < NF; ++fn) {
      int n = col0 + wn * NF * 16 + fn * 16 + g * 4;
      bb[fn] = *reinterpret_cast<const uint2*>(p.bias + (n < p.Cn ? n : 0));
    }
  }
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int m = row0 + wm * 64 + fm * 16 + l15;
    const bool mok = m < p.R;
    const size_t rowoff = (size_t)(mok ? m : 0) * p.ldc;
    if constexpr (F32OUT) {
      float* Cf = reinterpret_cast<float*>(p.C) + (size_t)blockIdx.z * p.R * p.ldc;
#pragma unroll
      for (int fn = 0; fn < NF; ++fn) {
        const int n = col0 + wn * NF * 16 + fn * 16 + g * 4;
        f32x4_t v = acc[fm][fn];
        if (mok && n < p.Cn) *reinterpret_cast<float4*>(Cf + rowoff + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else if (p.gu) {
      // fused SwiGLU backward: acc = d(act)[m][c..c+3]; gate at gu[m][(c/32)*64 + c%32], up 32 columns later
      uint2 gg[NF], uu[NF];
      bf16_t* grow = p.gu + (size_t)(mok ? m : 0) * (2 * p.Cn);
#pragma unroll
      for (int fn = 0; fn < NF; ++fn) {
        int c = col0 + wn * NF * 16 + fn * 16 + g * 4;
        c = c < p.Cn ? c : 0;
        const bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
        gg[fn] = *reinterpret_cast<const uint2*>(gp);
        uu[fn] = *reinterpret_cast<const uint2*>(gp + 32);
      }
#pragma unroll
      for (int fn = 0; fn < NF; ++fn) {
        const int c = col0 + wn * NF * 16 + fn * 16 + g * 4;
        f32x4_t d = acc[fm][fn];
        const float gv[4] = {__uint_as_float(gg[fn].x << 16), __uint_as_float(gg[fn].x & 0xffff0000u),
                             __uint_as_float(gg[fn].y << 16), __uint_as_float(gg[fn].y & 0xffff0000u)};
        const float uv[4] = {__uint_as_float(uu[fn].x << 16), __uint_as_float(uu[fn].x & 0xffff0000u),
                             __uint_as_float(uu[fn].y << 16), __uint_as_float(uu[fn].y & 0xffff0000u)};
        float dg[4], du[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sg = fast_sigmoid(gv[r]);
          du[r] = d[r] * gv[r] * sg;
          dg[r] = d[r] * uv[r] * sg * (1.f + gv[r] * (1.f - sg));
        }
        if (mok && c < p.Cn) {
          bf16_t* gp = grow + (c >> 5) * 64 + (c & 31);
          uint2 o1, o2;
          o1.x = pack_bf16x2(dg[0], dg[1]); o1.y = pack_bf16x2(dg[2], dg[3]);
          o2.x = pack_bf16x2(du[0], du[1]); o2.y = pack_bf16x2(du[2], du[3]);
          *reinterpret_cast<uint2*>(gp) = o1;
          *reinterpret_cast<uint2*>(gp + 32) = o2;
        }
      }
    } else {
      uint2 rr[4];
      if (p.resid) {
#pragma unroll
        for (int fn = 0; fn < NF; ++fn) {
          int n = col0 + wn * NF * 16 + fn * 16 + g * 4;
          rr[fn] = *reinterpret_cast<const uint2*>(p.resid + rowoff + (n < p.Cn ? n : 0));
        }
      }
      f32x4_t v[NF];
#pragma unroll
      for (int fn = 0; fn < NF; ++fn) {
        v[fn] = acc[fm][fn];
        if (p.bias) {
          v[fn][0] += __uint_as_float(bb[fn].x << 16); v[fn][1] += __uint_as_float(bb[fn].x & 0xffff0000u);
          v[fn][2] += __uint_as_float(bb[fn].y << 16); v[fn][3] += __uint_as_float(bb[fn].y & 0xffff0000u);
        }
        if (p.resid) {
          v[fn][0] += __uint_as_float(rr[fn].x << 16); v[fn][1] += __uint_as_float(rr[fn].x & 0xffff0000u);
          v[fn][2] += __uint_as_float(rr[fn].y << 16); v[fn][3] += __uint_as_float(rr[fn].y & 0xffff0000u);
        }
      }
      if constexpr (NF == 4) {
        // fused RoPE: the wave's 64 columns are one head; fragments fn and fn+2 hold d and d+32
        if (p.rope_cos && mok && ((col0 + wn * 64) >> 6) < p.rope_heads) {
          const bool qh = ((col0 + wn * 64) >> 6) < p.rope_q_heads;
          const float* ct = qh ? p.rope_cos_q : p.rope_cos;
          const float* stb = qh ? p.rope_sin_q : p.rope_sin;
#pragma unroll
          for (int fn = 0; fn < 2; ++fn) {
            const float4 c4 = *reinterpret_cast<const float4*>(ct + (size_t)m * 32 + fn * 16 + g * 4);
            const float4 s4 = *reinterpret_cast<const float4*>(stb + (size_t)m * 32 + fn * 16 + g * 4);
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x1 = v[fn][r], x2 = v[fn + 2][r];
              v[fn][r] = x1 * cc[r] - x2 * ss[r];
              v[fn + 2][r] = x2 * cc[r] + x1 * ss[r];
            }
          }
        }
      }
#pragma unroll
      for (int fn = 0; fn < NF; ++fn) {
        const int n = col0 + wn * NF * 16 + fn * 16 + g * 4;
        uint2 o;
        o.x = pack_bf16x2(v[fn][0], v[fn][1]);
        o.y = pack_bf16x2(v[fn][2], v[fn][3]);
        if (mok && n < p.Cn) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + rowoff + n) = o;
      }
      if constexpr (NF == 4) {
        // fused SwiGLU: this wave's 64 columns are [32 gate | 32 up]; fragments fn and fn+2 of a lane
        // hold gate and up of the same 4 activation columns
        if (p.act && mok) {
#pragma unroll
          for (int fn = 0; fn < 2; ++fn) {
            const int ac = (col0 + wn * 64) / 2 + fn * 16 + g * 4;
            f32x4_t gt = acc[fm][fn], up = acc[fm][fn + 2];
            float a0 = gt[0] * fast_sigmoid(gt[0]) * up[0], a1 = gt[1] * fast_sigmoid(gt[1]) * up[1];
            float a2 = gt[2] * fast_sigmoid(gt[2]) * up[2], a3 = gt[3] * fast_sigmoid(gt[3]) * up[3];
            uint2 o;
            o.x = pack_bf16x2(a0, a1);
            o.y = pack_bf16x2(a2, a3);
            *reinterpret_cast<uint2*>(p.act + (size_t)m * (p.Cn / 2) + ac) = o;
          }
        }
      }
    }
  }
}

// ---- 256 x 256 tiles, 8 waves, 8-phase schedule ------------------------------------------------------------
// For the wide-N projections (gate|up forward, down-proj dgrad). Eight waves as 2 (rows) x 4 (columns), each
// owning 128 x 64 of the output (acc = 128 VGPRs); one block per CU (128 KB of LDS: two K-tile buffers of four
// 16 KB half-tiles). A K-tile is consumed in four phases, one 64 x 32 quadrant of the wave's output each
// (16 MFMAs): the half-tiles are cut so that a phase needs at most one new one -
//   Amq_h = rows {64h .. 64h+63} of BOTH wave rows,  Bnq_h = columns {32h .. 32h+31} of ALL FOUR wave columns,
//   phase 1: Amq0 + Bnq0 -> quadrant (0,0); 2: Bnq1 -> (0,1); 3: Amq1 -> (1,1); 4: Bnq0 again (registers) -> (1,0)
// and every phase issues the DMA of ONE half-tile of the next K-tile (same order), so three half-tiles are
// always in flight and the counted vmcnt never drains. Each phase is
//   [issue DMA | ds_read the new fragments | counted vmcnt] barrier [16 MFMA at raised priority] barrier
// and the second wave row runs one barrier behind the first: while one wave of a SIMD is in its MFMA section the
// other one is loading, by construction rather than by chance. A wait that retires a half-tile sits in the
// phase BEFORE the one that reads it (the barrier in between publishes every wave's DMA).
// Column-tile rows are in the perm64 order, so the epilogue is epilogue8 on the two 64-row halves.
template <bool LAST>
SLAM_DEVICE void wait_ph(int which) {
  // outstanding half-tiles (2 DMAs each) allowed after the wait: 2 in steady state, fewer on the last K-tile
  // (SLAM_PROBE_VMCNT: a stricter steady-state count for the prefetch-depth probe of tools/probes/depth_probe.sh)
#if !defined(SLAM_PROBES) || !defined(SLAM_PROBE_VMCNT)
#undef SLAM_PROBE_VMCNT
#define SLAM_PROBE_VMCNT 4
#endif
  if (which == 0) { if (LAST) wait_vmcnt<2>(); else wait_vmcnt<SLAM_PROBE_VMCNT>(); }
  else if (which == 1) { if (LAST) wait_vmcnt<0>(); else wait_vmcnt<SLAM_PROBE_VMCNT>(); }
  else { if (!LAST) wait_vmcnt<SLAM_PROBE_VMCNT>(); }
}
SLAM_DEVICE void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// (measured against this schedule: one barrier per phase without the wave-row stagger +3 % time, no priority raise +8 %;
//  round 2: starting groups of first-round blocks a fraction of a tile period late, to de-synchronise the epilogue store
//  bursts of the 256 resident blocks, moves the fused gate|up launch by <= 4 % in isolation and the step by nothing)
// PERSIST: gridDim.x (a multiple of 8, <= the CU count) blocks walk the tile list instead of one block per tile - XCD x
// keeps its contiguous range of tile ids, slot s of the XCD takes ids s, s + slots, ... (the co-running set of a round is
// the one the dispatcher produces for the one-tile-per-block grid). The DMA stream does not stop at a tile boundary: the
// last K-tile of a tile issues the first K-tile of the next one (steady-state waits), the block then drains ITS loads
// (vmcnt counts stores and loads in one in-order queue: a counted wait after the epilogue would wait for the stores),
// stores the finished tile and continues with a K-tile whose first two phases need no wait. Dispatch, address set-up and
// the first-load latency of a tile disappear behind the previous tile's epilogue.
// MF32 (round 5): the same schedule on v_mfma_f32_32x32x16_bf16 - a phase is 2 (32-row blocks) x 1 (32-column block) x 4
// K-steps of 16 = 8 MFMAs of 32 cycles instead of 16 of ~17; the same 12 ds_read_b128 per phase (a fragment is now 32 rows
// x 16 k instead of 16 x 32), the same LDS images; column-tile rows in the perm32 order, epilogue32.
// ROLES (round 6, PERSIST && !MF32 only, "gemm_256_roles"): loads and stores share ONE in-order vmcnt per wave, so a wave that
// has just stored its tile cannot use a counted wait for the next K-tiles until those stores have drained - the 256 resident
// blocks sit through their store bursts together (profiles/r5_experiments/README.md section 1: +20 ... +72 % cycles). The
// two wave rows take different jobs instead (one wave of each row per SIMD):
//   wave row 0 = LOADERS: issue EVERY LDS-DMA of the block (4 per half-tile instead of 2) and do every counted wait - their
//                vmcnt only ever holds loads; they never store to memory: a finished quadrant goes to LDS as 16-byte records
//                (quad_emit<true>) into the K-tile buffer that is free at the tile boundary;
//   wave row 1 = STORERS: issue no load in the K loop and execute no vmcnt wait there (the barriers publish the loaders'
//                waits, as they always did for the other waves' slices); they store their own quadrants and the loaders'
//                records (quad_drain) and run on into the next tile's K loop with the stores in flight.
// Four barriers per tile on top of the schedule (records visible / staging region free, per 64-row quadrant).
template <bool PERSIST, bool MF32, bool ROLES = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_256_kernel(GemmArgs p) {
  static_assert(!ROLES || (PERSIST && !MF32), "role split: persistent 16x16x32 kernel only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HT = 128 * 128;      // half-tile bytes: 128 rows x 128 B
  constexpr int KT = 4 * HT;         // K-tile buffer
  constexpr int NDMA = ROLES ? 4 : 2;  // LDS-DMA instructions per half-tile and issuing wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, g = lane >> 4;
  const int l31 = lane & 31, hh = lane >> 5;
  const int nblk = p.tiles_r * p.tiles_c;
  int nid, nid_end, nid_step;
  {
    int id = blockIdx.x, xcd = id & 7, idx = id >> 3;
    int q = nblk >> 3, r = nblk & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    nid = base + idx;
    nid_end = PERSIST ? base + q + (xcd < r ? 1 : 0) : nid + 1;
    nid_step = PERSIST ? (int)(gridDim.x >> 3) : 1;
    if (PERSIST && p.stagger_ticks > 0) {
      // The 256 resident blocks run K loop (MFMA-bound) and epilogue (HBM-bound: the fused SwiGLU backward moves 512 KB per
      // tile) in lockstep. Slots whose tile list is one shorter than the longest have a whole tile period of slack: they
      // start late, so their epilogues fall under the other blocks' K loops at no cost in makespan.
      const int mine = (nid_end - nid + nid_step - 1) / nid_step, longest = (nid_end - base + nid_step - 1) / nid_step;
      // cohorts > 1 (round 5 probe): EVERY slot is delayed by its cohort's offset - the long tile lists sit in the low
      // slots, i.e. in the early cohorts
      const int delay = p.cohorts > 1 ? (idx * p.cohorts / nid_step) * p.stagger_ticks : (mine < longest ? p.stagger_ticks : 0);
      if (delay > 0) {
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < (uint64_t)delay) __builtin_amdgcn_s_sleep(32);
      }
    }
  }
  auto tile_origin = [&](int id, int& r0, int& c0) {
    // bands of GR row tiles; inside a band, groups of GC column tiles (GC = all columns by default); inside a group column-major:
    // the tiles in flight on one XCD share GR row panels and a few column panels, and with GC < tiles_c an XCD's whole range of
    // tile ids touches only GC column panels of the weight (2-D partition of the tile space over the XCD-private L2s)
    const int GR = p.group_rows > 0 ? p.group_rows : 1;
    const int GC = p.group_cols > 0 ? min(p.group_cols, p.tiles_c) : p.tiles_c;
    const int per_band = GR * p.tiles_c;
    const int band = id / per_band, in = id - band * per_band;
    const int rows_here = min(GR, p.tiles_r - band * GR);
    const int cg = in / (rows_here * GC), in2 = in - cg * rows_here * GC;
    const int tc_ = in2 / rows_here;
    r0 = (band * GR + in2 - tc_ * rows_here) * 256;
    c0 = (cg * GC + tc_) * 256;
  };
  int row0, col0;
  tile_origin(nid, row0, col0);
  const int nk = p.Kc / BK;
  const uint32_t lds0 = lds_addr(smem);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool loader = !ROLES || wv < 4;  // wave-uniform (SGPR)
  // DMA source offsets, NDMA chunks per lane per half-tile; half-tile order in a buffer: Amq0 | Bnq0 | Bnq1 | Amq1
  uint32_t vo[4][NDMA];
  // fragment addresses inside a half-tile: byte offsets of row (wave block + f*16 + l15), without the chunk term
  int ka, offA[4], offB[2];
  // per-lane constants of the K loop. A persistent block computes them again after every epilogue (from an opaque copy
  // of the thread id, so that the compiler cannot keep the first set alive): they stay out of the epilogue's live range
  auto lane_setup = [&](int t_) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int P = ROLES ? i * 256 + (t_ & 255) : i * 512 + t_, r = P >> 3, c = (P & 7) ^ lds_swz_key(r);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int grow = (r >> 6) * 128 + h * 64 + (r & 63);          // relative to the tile origin: 32-bit offsets must not
        const int gcol = (r >> 5) * 64 + (MF32 ? h * 32 + perm32(r & 31) : perm64(h * 32 + (r & 31)));   // span the matrix (dlogits [16384][152320] is 4.99 GB)
        vo[h ? 3 : 0][i] = (uint32_t)(((size_t)grow * p.lda + c * 8) * sizeof(bf16_t));
        vo[1 + h][i] = (uint32_t)(((size_t)gcol * p.ldb + c * 8) * sizeof(bf16_t));
      }
    }
    if constexpr (MF32) {
      // 32x32x16 fragments: row (wave block + b*32 + l31), K-step s: chunk (2s + hh) ^ key(row), key = ka ^ (row block >> 4)
      const int l31_ = t_ & 31;
      ka = ((l31_ >> 1) ^ (l31_ >> 4)) & 7;
#pragma unroll
      for (int b = 0; b < 2; ++b) offA[b] = (wr * 64 + b * 32 + l31_) * 128;
      offB[0] = (wc * 32 + l31_) * 128;
    } else {
      const int l15_ = t_ & 15;
      ka = (l15_ >> 1) & 7;
#pragma unroll
      for (int f = 0; f < 4; ++f) offA[f] = (wr * 64 + f * 16 + l15_) * 128;
#pragma unroll
      for (int f = 0; f < 2; ++f) offB[f] = (wc * 32 + f * 16 + l15_) * 128;
    }
  };
  lane_setup(tid);
  // h: position in the buffer (0 Amq0, 1 Bnq0, 2 Bnq1, 3 Amq1); ta / tb: the K-tile's origin in A / B; par: buffer
  auto issue_half = [&](int h, const bf16_t* ta, const bf16_t* tb, int par) {
    if (ROLES && !loader) return;
    const bf16_t* base = (h == 0 || h == 3) ? ta : tb;
    const uint32_t dst = lds0 + (uint32_t)(par * KT + h * HT) + (uint32_t)wv * 1024u;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) glds16_sv(base, vo[h][i], __builtin_amdgcn_readfirstlane(dst + (uint32_t)(i * (ROLES ? 4096 : 8192))));
  };
  // counted waits: the loaders' counts are in units of their 4 DMAs per half-tile; storers never wait on vmcnt in the K loop
  auto wait_phase = [&](auto last_tag, int which) {
    constexpr bool LAST = decltype(last_tag)::value;
    if constexpr (ROLES) {
      if (!loader) return;
      if (which == 0) { if (LAST) wait_vmcnt<4>(); else wait_vmcnt<8>(); }
      else if (which == 1) { if (LAST) wait_vmcnt<0>(); else wait_vmcnt<8>(); }
      else { if (!LAST) wait_vmcnt<8>(); }
    } else {
      wait_ph<LAST>(which);
    }
  };

  f32x4_t acc[2][4][4];      // 16x16x32 form (dead in the MF32 instantiation)
  f32x16_t acc32[2][2][2];   // 32x32x16 form: [row quadrant mq][32-row block][column quadrant nq]
  auto zero_acc = [&]() {
    if constexpr (MF32) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[a][i][j][e] = 0.f;
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
  };
  zero_acc();

  uint4 afr[2][4], bfr[2][2][2];  // 16x16x32: [kk][fm], [nq][kk][fn];  32x32x16: A K-step s, block b at afr[s >> 1][2 (s & 1) + b], B at bfr[nq][s >> 1][s & 1]
  auto read_A = [&](const char* half) {
    if constexpr (MF32) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          afr[ks >> 1][2 * (ks & 1) + b] = *reinterpret_cast<const uint4*>(half + offA[b] + ((((2 * ks + hh) ^ ka ^ (wr * 4 + b * 2)) & 7) << 4));
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int f = 0; f < 4; ++f)
          afr[kk][f] = *reinterpret_cast<const uint4*>(half + offA[f] + ((((g + 4 * kk) ^ ka ^ (wr * 4 + f)) & 7) << 4));
    }
  };
  auto read_B = [&](const char* half, int nq) {
    if constexpr (MF32) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        bfr[nq][ks >> 1][ks & 1] = *reinterpret_cast<const uint4*>(half + offB[0] + ((((2 * ks + hh) ^ ka ^ (wc * 2)) & 7) << 4));
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          bfr[nq][kk][f] = *reinterpret_cast<const uint4*>(half + offB[f] + ((((g + 4 * kk) ^ ka ^ (wc * 2 + f)) & 7) << 4));
    }
  };
  auto barrier_b = [&]() { raw_barrier(); };
  auto mma = [&](int mq, int nq) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (MF32) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc32[mq][b][nq] = mfma32(bfr[nq][ks >> 1][ks & 1], afr[ks >> 1][2 * (ks & 1) + b], acc32[mq][b][nq]);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
          for (int fn = 0; fn < 2; ++fn)
            acc[mq][fm][nq * 2 + fn] = mfma16(bfr[nq][kk][fn], afr[kk][fm], acc[mq][fm][nq * 2 + fn]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // 64 rows x the wave's 64 columns per call: row quadrant mq of the wave
  auto store_quadrant = [&](auto lean_tag, int mq, int a_, int b_) {
    constexpr bool LEAN = decltype(lean_tag)::value;
    if constexpr (MF32) epilogue32<LEAN>(p, acc32[mq], row0 + wr * 128 + mq * 64, col0 + wc * 64, a_, b_);
    else epilogue8<LEAN>(p, acc[mq], row0 + wr * 128 + mq * 64, col0, 0, wc, a_, b_);
  };
  // MODE 0: steady state (issues the K-tile at na / nb into the other buffer); 1: last K-tile of the block (nothing to
  // issue, waits drain); 2: first K-tile after a tile boundary of a persistent block - its own half-tiles were drained
  // before the epilogue, and a counted wait in phases 1 / 2 would wait for the epilogue's stores
  auto ktile = [&](int par, auto mode_tag, const bf16_t* na, const bf16_t* nb) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool LAST = MODE == 1;
    const char* buf = smem + par * KT;
    // phase 1
    if (!LAST) issue_half(0, na, nb, par ^ 1);
    read_A(buf);
    read_B(buf + HT, 0);
    if (MODE != 2) wait_phase(std::integral_constant<bool, LAST>{}, 0);  // Bnq1(t) landed -> read in phase 2
    raw_barrier();
    mma(0, 0);
    barrier_b();
    // phase 2
    if (!LAST) issue_half(1, na, nb, par ^ 1);
    read_B(buf + 2 * HT, 1);
    if (MODE != 2) wait_phase(std::integral_constant<bool, LAST>{}, 1);  // Amq1(t) landed -> read in phase 3
    raw_barrier();
    mma(0, 1);
    barrier_b();
    // phase 3
    if (!LAST) issue_half(2, na, nb, par ^ 1);
    read_A(buf + 3 * HT);
    raw_barrier();
    mma(1, 1);
    barrier_b();
    // phase 4
    if (!LAST) issue_half(3, na, nb, par ^ 1);
    wait_phase(std::integral_constant<bool, LAST>{}, 2);  // Amq0(t+1), Bnq0(t+1) landed -> read in phase 1 of the next K-tile
    raw_barrier();
    mma(1, 0);
    barrier_b();
  };
  using steady_t = std::integral_constant<int, 0>;
  using last_t = std::integral_constant<int, 1>;
  using first_t = std::integral_constant<int, 2>;

  // prologue: K-tile 0 in the order it is needed; phase 1 needs the first two half-tiles
  const bf16_t* ta = p.A + (size_t)row0 * p.lda;
  const bf16_t* tb = p.B + (size_t)col0 * p.ldb;
  issue_half(0, ta, tb, 0);
  issue_half(1, ta, tb, 0);
  issue_half(2, ta, tb, 0);
  issue_half(3, ta, tb, 0);
  if constexpr (!PERSIST) {
    wait_vmcnt<4>();
    raw_barrier();
    if (wr == 1) raw_barrier();  // second wave row: one barrier behind from here on
    for (int t = 0; t + 1 < nk; ++t) ktile(t & 1, steady_t{}, ta + (size_t)(t + 1) * BK, tb + (size_t)(t + 1) * BK);
    ktile((nk - 1) & 1, last_t{}, nullptr, nullptr);
    if (wr == 0) raw_barrier();  // balance the barrier count
    store_quadrant(std::false_type{}, 0, MF32 ? l31 : l15, MF32 ? hh : g);
    store_quadrant(std::false_type{}, 1, MF32 ? l31 : l15, MF32 ? hh : g);
  } else {
    // one loop body for every tile (no variant diamonds: the accumulators keep their registers): the first K-tile of a
    // tile finds its half-tiles drained, the last one issues the next tile's first K-tile - or, on the block's last tile,
    // this tile's first K-tile once more (64 KB of L2 reads per block, never consumed)
    if (loader) wait_vmcnt<0>();
    raw_barrier();
    int par = 0;
    for (;;) {
      const int nnext = nid + nid_step;
      const bool has_next = nnext < nid_end;
      if (wr == 1) raw_barrier();  // second wave row: one barrier behind inside a tile
      ktile(par, first_t{}, ta + BK, tb + BK);
      par ^= 1;
      for (int t = 1; t + 1 < nk; ++t) {
        ktile(par, steady_t{}, ta + (size_t)(t + 1) * BK, tb + (size_t)(t + 1) * BK);
        par ^= 1;
      }
      int nrow0 = row0, ncol0 = col0;
      if (has_next) tile_origin(nnext, nrow0, ncol0);
      ta = p.A + (size_t)nrow0 * p.lda;
      tb = p.B + (size_t)ncol0 * p.ldb;
      ktile(par, steady_t{}, ta, tb);  // the DMA stream runs on into the next tile
      par ^= 1;
      if (wr == 0) raw_barrier();  // balance the barrier count: both wave rows leave the tile together
      if (loader) wait_vmcnt<0>();  // this wave's pieces of the next tile's first K-tile (see MODE 2)
      int l15e = MF32 ? l31 : l15, ge = MF32 ? hh : g;
      asm volatile("" : "+v"(l15e), "+v"(ge));  // keeps the epilogue's address arithmetic out of the K loop
      if constexpr (ROLES) {
        // staging: the K-tile buffer the last K-tile was read from (the next tile's first K-tile sits in the other one and the
        // first DMA into this one is issued after the last barrier below); 16 KB per wave column, record k of lane l at k KB + 16 l
        char* stg = smem + (par ^ 1) * KT + wc * 16384 + lane * 16;
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
          if (loader) {
            quad_emit<true>(p, acc[mq], row0 + mq * 64, col0, wc, l15e, ge, stg);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          } else {
            quad_emit<false>(p, acc[mq], row0 + 128 + mq * 64, col0, wc, l15e, ge, nullptr);  // own rows, beside the loaders' conversion
          }
          raw_barrier();  // the loaders' records are in LDS
          if (!loader) quad_drain(p, row0 + mq * 64, col0, wc, l15e, ge, stg);
          raw_barrier();  // the staging region is free again (next quadrant / the DMA of the next tile's second K-tile)
        }
      } else if (!MF32 && p.gu && p.batch_epilogue_loads) {
        if constexpr (!MF32) dswiglu_tile(p, acc, row0 + wr * 128, col0, wc, l15e, ge);
      } else {
        store_quadrant(std::true_type{}, 0, l15e, ge);
        store_quadrant(std::true_type{}, 1, l15e, ge);
      }
      if (!has_next) break;
      zero_acc();
      nid = nnext; row0 = nrow0; col0 = ncol0;
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      lane_setup(t_);
    }
  }
}

// ---- NT GEMM on 256 x 256 tiles, FOUR waves of 128 x 128 on v_mfma_f32_32x32x16_bf16, persistent blocks (round 5) -------
// The eight-wave kernel above reads 192 KB of fragments from LDS per 64-deep K-tile (24 KB per wave) and issues 128 MFMAs of
// the 16x16x32 form per wave. Here a wave owns 128 x 128 of the output - 4 x 4 blocks of 32 x 32, 256 accumulator registers
// pinned to the AGPR half of the unified file by the MFMA asm's "a" constraint - and a K-tile is four K-steps of 16:
// 8 ds_read_b128 (4 row-tile + 4 column-tile fragments of 32 rows x 16 k) per 16 MFMAs of 32 cycles: 128 KB of fragment reads
// per K-tile and 64 MFMA issues per wave instead of 128. One wave per SIMD has no neighbour to hide behind, so the overlap is
// inside the wave: while the matrix pipe runs K-step s the wave issues the reads of K-step s + 1 into the other fragment
// register set, and the LDS-DMA of later K-tiles, two instructions per MFMA row of four. Two whole-K-tile LDS buffers
// (A image | B image, 64 KB each), ONE barrier per K-tile, in front of K-step 3: K-tile t+1 has landed for every wave and
// every wave has read the last fragments of K-tile t, so K-step 3 reads K-step 0 of tile t+1 and starts the DMA of tile
// t+2 into the buffer of tile t (first DMA3 pieces; the rest follow in K-step 0 of tile t+1: a piece has 1.5 - 2.5 K-steps
// to land). Persistent blocks as in the eight-wave kernel: the K-tile sequence runs on across tile boundaries, the
// accumulators restart through the first K-step's MFMAs (C = 0 inline constant), the epilogue (epilogue32, LEAN) sits
// between the last K-step of a tile and the first of the next with K-tile 0 of the next tile already in registers / LDS.
template <int DMA3>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = 256 * 128;   // one operand image: 256 rows x 128 B
  constexpr int KT = 2 * IMG;      // K-tile buffer: A image | B image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  const int nblk = p.tiles_r * p.tiles_c;
  int nid, nid_end, nid_step;
  {
    int id = blockIdx.x, xcd = id & 7, idx = id >> 3;
    int q = nblk >> 3, r = nblk & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    nid = base + idx;
    nid_end = base + q + (xcd < r ? 1 : 0);
    nid_step = (int)(gridDim.x >> 3);
    if (p.stagger_ticks > 0) {  // blocks with a tile of slack start late: their store bursts fall under the others' K loops
      const int mine = (nid_end - nid + nid_step - 1) / nid_step, longest = (nid_end - base + nid_step - 1) / nid_step;
      if (mine < longest) {
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < (uint64_t)p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
      }
    }
  }
  if (nid >= nid_end) return;
  auto tile_origin = [&](int id, int& r0, int& c0) {
    // bands of GR row tiles; inside a band, groups of GC column tiles (GC = all columns by default); inside a group column-major:
    // the tiles in flight on one XCD share GR row panels and a few column panels, and with GC < tiles_c an XCD's whole range of
    // tile ids touches only GC column panels of the weight (2-D partition of the tile space over the XCD-private L2s)
    const int GR = p.group_rows > 0 ? p.group_rows : 1;
    const int GC = p.group_cols > 0 ? min(p.group_cols, p.tiles_c) : p.tiles_c;
    const int per_band = GR * p.tiles_c;
    const int band = id / per_band, in = id - band * per_band;
    const int rows_here = min(GR, p.tiles_r - band * GR);
    const int cg = in / (rows_here * GC), in2 = in - cg * rows_here * GC;
    const int tc_ = in2 / rows_here;
    r0 = (band * GR + in2 - tc_ * rows_here) * 256;
    c0 = (cg * GC + tc_) * 256;
  };
  int row0, col0;
  tile_origin(nid, row0, col0);
  const int nk = p.Kc / BK;
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t wv1k = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
  // DMA: 8 + 8 pieces of 1 KB per wave per K-tile; LDS chunk P = i * 256 + tid is row P >> 3, slot P & 7 and holds the source
  // chunk slot ^ key(row). B rows are in the perm32 order of each 32-row block (epilogue32's 16 consecutive columns per lane).
  uint32_t voa[8], vob[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int P = i * 256 + tid, r = P >> 3, c = (P & 7) ^ lds_swz_key(r);
    const int gcol = (r & ~31) + perm32(r & 31);
    voa[i] = (uint32_t)(((size_t)r * p.lda + c * 8) * sizeof(bf16_t));      // relative to the tile origin
    vob[i] = (uint32_t)(((size_t)gcol * p.ldb + c * 8) * sizeof(bf16_t));
  }
  // piece q of a K-tile (q < 8: A image, else B image) from the K-tile's origins (ka, kb) into the buffer at `buf`
  auto issue_piece = [&](int q, const bf16_t* ka_, const bf16_t* kb_, uint32_t buf) __attribute__((always_inline)) {
    if (q < 8) glds16_m0(ka_, voa[q], buf + wv1k + (uint32_t)(q * 4096));
    else glds16_m0(kb_, vob[q - 8], buf + (uint32_t)IMG + wv1k + (uint32_t)((q - 8) * 4096));
  };
  // fragment (32-row block b of the wave's 128, K-step s): row w*128 + b*32 + l31, chunk (2s + hh) ^ key(row) =
  // x ^ 2(s ^ b) for the lane constant x = hh ^ ((l31>>1) ^ (l31>>4)) & 7: four lane offsets per operand serve every fragment
  int lo_a[4], lo_b[4];
  {
    const int x = hh ^ (((l31 >> 1) ^ (l31 >> 4)) & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo_a[j] = (wr * 128 + l31) * 128 + (((x ^ (2 * j)) & 7) << 4);
      lo_b[j] = IMG + (wc * 128 + l31) * 128 + (((x ^ (2 * j)) & 7) << 4);
    }
  }
  f32x16_t acc[2][2][2][2];  // [64-row half][64-column half][32-row block][32-column block]: four epilogue32 pieces
  u32x4_t fa[2][4], fb[2][4];  // [K-step parity][block]
  typedef __attribute__((address_space(3))) const u32x4_t* lds_v4_t;
  auto rd_a = [&](uint32_t buf, int ks, int b) __attribute__((always_inline)) {
    fa[ks & 1][b] = *(lds_v4_t)(uintptr_t)(buf + (uint32_t)lo_a[(ks ^ b) & 3] + (uint32_t)(b * 4096));
  };
  auto rd_b = [&](uint32_t buf, int ks, int b) __attribute__((always_inline)) {
    fb[ks & 1][b] = *(lds_v4_t)(uintptr_t)(buf + (uint32_t)lo_b[(ks ^ b) & 3] + (uint32_t)(b * 4096));
  };
  // the fragments of K-step ks, two per MFMA row: the next K-step's first row needs every B fragment and A block 0
  auto rd_row = [&](uint32_t buf, int ks, int row) __attribute__((always_inline)) {
    if (row == 0) { rd_b(buf, ks, 0); rd_b(buf, ks, 1); }
    else if (row == 1) { rd_b(buf, ks, 2); rd_b(buf, ks, 3); }
    else if (row == 2) { rd_a(buf, ks, 0); rd_a(buf, ks, 1); }
    else { rd_a(buf, ks, 2); rd_a(buf, ks, 3); }
  };
  auto mma_row = [&](int ks, int rb) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[rb >> 1][nb >> 1][rb & 1][nb & 1]) : "v"(fb[ks & 1][nb]), "v"(fa[ks & 1][rb]));
  };
  auto mma_row_zero = [&](int ks, int rb) __attribute__((always_inline)) {  // C = 0: the accumulators restart
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[rb >> 1][nb >> 1][rb & 1][nb & 1]) : "v"(fb[ks & 1][nb]), "v"(fa[ks & 1][rb]));
  };
  // One K-tile. buf holds it, nbuf the next one. (a1, b1): origins of K-tile t+1 (its pieces DMA3..15 are issued in K-step
  // 0 into nbuf), (a2, b2): origins of K-tile t+2 (pieces 0..DMA3-1 issued in K-step 3 into buf). FIRST: first K-tile of an
  // output tile (the accumulators restart).
  auto ktile = [&](uint32_t buf, uint32_t nbuf, const bf16_t* a1, const bf16_t* b1, const bf16_t* a2, const bf16_t* b2,
                   auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int PER0 = (16 - DMA3 + 3) / 4, PER3 = (DMA3 + 3) / 4;  // pieces per MFMA row
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {  // K-step 0 under the reads of K-step 1
      rd_row(buf, 1, rb);
#pragma unroll
      for (int j = 0; j < PER0; ++j)
        if (DMA3 + rb * PER0 + j < 16) issue_piece(DMA3 + rb * PER0 + j, a1, b1, nbuf);
      if constexpr (FIRST) mma_row_zero(0, rb);
      else mma_row(0, rb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {  // K-step 1 under the reads of K-step 2
      rd_row(buf, 2, rb);
      mma_row(1, rb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {  // K-step 2 under the reads of K-step 3
      rd_row(buf, 3, rb);
      mma_row(2, rb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // K-tile t+1 has landed (this wave's pieces; the barrier publishes everyone's), every wave has read K-tile t
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    raw_barrier();
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {  // K-step 3 under the reads of K-step 0 of the next K-tile
      rd_row(nbuf, 0, rb);
#pragma unroll
      for (int j = 0; j < PER3; ++j)
        if (rb * PER3 + j < DMA3) issue_piece(rb * PER3 + j, a2, b2, buf);
      mma_row(3, rb);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: K-tile 0 of the first tile, whole; then its K-step 0 fragments and the first pieces of K-tile 1
  const bf16_t* ta = p.A + (size_t)row0 * p.lda;
  const bf16_t* tb = p.B + (size_t)col0 * p.ldb;
#pragma unroll
  for (int q = 0; q < 16; ++q) issue_piece(q, ta, tb, lds0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) rd_row(lds0, 0, rb);
#pragma unroll
  for (int q = 0; q < DMA3; ++q) issue_piece(q, ta + BK, tb + BK, lds0 + (uint32_t)KT);
  uint32_t par = 0;
  for (;;) {
    const int nnext = nid + nid_step;
    const bool has_next = nnext < nid_end;
    int nrow0 = row0, ncol0 = col0;
    if (has_next) tile_origin(nnext, nrow0, ncol0);
    // the tile the DMA stream runs on into (the block's last tile: this tile once more, never consumed)
    const bf16_t* na = p.A + (size_t)nrow0 * p.lda;
    const bf16_t* nb_ = p.B + (size_t)ncol0 * p.ldb;
    // K-tile 0 (nk >= 2: K-tile 1 is this tile's; K-tile 2 is this tile's or the next tile's first)
    {
      const uint32_t buf = lds0 + par * (uint32_t)KT, nbuf = lds0 + (par ^ 1u) * (uint32_t)KT;
      const bool in2 = 2 < nk;
      ktile(buf, nbuf, ta + BK, tb + BK, in2 ? ta + 2 * BK : na, in2 ? tb + 2 * BK : nb_, std::true_type{});
      par ^= 1u;
    }
    for (int t = 1; t < nk; ++t) {
      const uint32_t buf = lds0 + par * (uint32_t)KT, nbuf = lds0 + (par ^ 1u) * (uint32_t)KT;
      const int u1 = t + 1, u2 = t + 2;
      const bf16_t* a1 = u1 < nk ? ta + (size_t)u1 * BK : na + (size_t)(u1 - nk) * BK;
      const bf16_t* b1 = u1 < nk ? tb + (size_t)u1 * BK : nb_ + (size_t)(u1 - nk) * BK;
      const bf16_t* a2 = u2 < nk ? ta + (size_t)u2 * BK : na + (size_t)(u2 - nk) * BK;
      const bf16_t* b2 = u2 < nk ? tb + (size_t)u2 * BK : nb_ + (size_t)(u2 - nk) * BK;
      ktile(buf, nbuf, a1, b1, a2, b2, std::false_type{});
      par ^= 1u;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs retire before the epilogue reads the accumulators
    {
      int l31e = l31, he = hh;
      asm volatile("" : "+v"(l31e), "+v"(he));  // keeps the epilogue's address arithmetic out of the K loop
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          epilogue32<true>(p, acc[a][b], row0 + wr * 128 + a * 64, col0 + wc * 128 + b * 64, l31e, he);
    }
    if (!has_next) break;
    nid = nnext; row0 = nrow0; col0 = ncol0;
    ta = na; tb = nb_;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-on pieces of the last tile must not land after the block has left
}

// ---- wgrad with balanced K-splitting ----------------------------------------------------------------
// dW[R][Cn] (fp32) (+)= A^T B with A [Kc][R], B [Kc][Cn] (both stored contraction-major), for the
// DMA-eligible shapes. Equal K-splits of every tile leave the 512 block slots (2 per CU) badly filled when
// tiles x splits is just above a multiple of 512 (gate|up weight: 532 tiles x 2 = 1064 blocks = three rounds
// of work for 2.08 rounds of blocks). Here the tiles are cut in two groups inside ONE launch:
//   A: the first T_A tiles, S_A pieces each, T_A*S_A a multiple of 512 (or <= 512): full rounds whose
//      co-running blocks walk the contraction in lockstep and share operand slices in L2;
//   B: the remaining T_B < 512/S_A tiles cut into many short pieces (S_B = up to 16) that fill the slots
//      once more for a fraction of a round.
// (A stream-K run-length decomposition balances even better on paper but skews the K positions of
//  co-running blocks: no L2 sharing, 2.2 GB of operand traffic per launch, -40 %.)
// Pieces of an unsplit tile accumulate straight into dW; split tiles write fp32 partial tiles to slabs and
// reduce_bal_kernel adds them in piece order: no atomics, the same bits every run.
struct BalArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* dW;
  float* wsA;  // [S_A][R][Cn]            (S_A > 1)
  float* wsB;  // [S_B][T_B][128 x 128]
  int lda, ldb, ldc;
  int tiles_r, tiles_c, KS, group_rows;
  int T_A, S_A, per_A, T_B, S_B, per_B, nA;
  int accumulate;
  size_t slab_stride;
  bf16_t* img;  // nullable: bf16 image of dW (same indexing), written together with every FINAL value of dW
  float* sumsq;      // nullable: GradSink slots of the GEMM kernel's blocks ...
  float* sumsq_red;  // ... and of reduce_bal_kernel's (y-major)
  int img_only;      // final values go to img only (GradSink)
};
SLAM_DEVICE void bal_tile_rc(int t, int tiles_r, int tiles_c, int group_rows, int& tr_, int& tc_) {
  const int GR = group_rows > 0 ? group_rows : 1;
  const int per_group = GR * tiles_c;
  const int grp = t / per_group, in = t - grp * per_group;
  const int rows_here = min(GR, tiles_r - grp * GR);
  tc_ = in / rows_here;
  tr_ = grp * GR + in - tc_ * rows_here;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_bal_kernel(BalArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  int t, ks0, ks1, stride;
  float* dst;
  bool direct = false;
  {
    int b = blockIdx.x;
    if (b < p.nA) {
      const int z = b / p.T_A, idx = b - z * p.T_A;
      // block b runs on XCD b % 8: give each XCD a contiguous run of tiles (T_A is a multiple of 8 when split)
      const int xcd = idx & 7, i8 = idx >> 3, q = p.T_A >> 3, r = p.T_A & 7;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i8;
      ks0 = z * p.per_A;
      ks1 = min(p.KS, ks0 + p.per_A);
      int tr_, tc_;
      bal_tile_rc(t, p.tiles_r, p.tiles_c, p.group_rows, tr_, tc_);
      direct = p.S_A == 1;
      dst = (direct ? p.dW : p.wsA + (size_t)z * p.slab_stride) + (size_t)tr_ * 128 * p.ldc + tc_ * 128;
      stride = p.ldc;
    } else {
      b -= p.nA;
      const int z = b / p.T_B, idx = b - z * p.T_B;
      t = p.T_A + idx;
      ks0 = z * p.per_B;
      ks1 = min(p.KS, ks0 + p.per_B);
      dst = p.wsB + ((size_t)z * p.T_B + idx) * (128 * 128);
      stride = 128;
    }
  }
  int tr_, tc_;
  bal_tile_rc(t, p.tiles_r, p.tiles_c, p.group_rows, tr_, tc_);
  const int nk = ks1 - ks0;
  const int trk = (l15 >> 2) | ((g & 1) << 2);
  const int tr_lane = (g * 8 + (l15 >> 2)) * 256 + (l15 & 3) * 8;
  const uint32_t lds0 = lds_addr(smem);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  uint32_t voa[4], vob[4];
  glds_offsets_tr<256, 128>(p.lda, tr_ * 128, tid, voa);
  glds_offsets_tr<256, 128>(p.ldb, tc_ * 128, tid, vob);
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int k) {
    const size_t k0 = (size_t)(ks0 + k) * BK;
    const uint32_t st = lds0 + (uint32_t)((k & 1) * STAGE_BYTES);
    glds_tile<256, 128>(p.A + k0 * p.lda, voa, wv, st);
    glds_tile<256, 128>(p.B + k0 * p.ldb, vob, wv, st + TILE_BYTES);
  };
  if (nk > 0) issue(0);
  for (int k = 0; k < nk; ++k) {
    wait_vmcnt<0>();
    __syncthreads();
    if (k + 1 < nk) issue(k + 1);
    const char* At = smem + (k & 1) * STAGE_BYTES;
    const char* Bt = At + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[4], bf[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const char* pa = Bt + tr_lane + kk * 32 * 256 + (((wn * 4 + f) ^ trk) << 5);
        uint2 a0 = lds_tr_read(pa), a1 = lds_tr_read(pa + 4 * 256);
        af[f] = make_uint4(a0.x, a0.y, a1.x, a1.y);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const char* pbp = At + tr_lane + kk * 32 * 256 + (((wm * 4 + f) ^ trk) << 5);
        uint2 b0 = lds_tr_read(pbp), b1 = lds_tr_read(pbp + 4 * 256);
        bf[f] = make_uint4(b0.x, b0.y, b1.x, b1.y);
      }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = mfma16(af[fn], bf[fm], acc[fm][fn]);
    }
  }
  const bool add = direct && p.accumulate;
  bf16_t* const imgt = (direct && p.img) ? p.img + (dst - p.dW) : nullptr;  // unsplit tile: this store is the final value
  const bool st32 = !(direct && p.img_only);  // a final value that is kept in bf16 only
  float ss = 0.f;
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const size_t ro = (size_t)(wm * 64 + fm * 16 + l15) * stride + wn * 64 + g * 4;
    float* row = dst + ro;
    float4 old[4];
    if (add) {
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) old[fn] = *reinterpret_cast<const float4*>(row + fn * 16);
    }
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      f32x4_t v = acc[fm][fn];
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (add) { o.x += old[fn].x; o.y += old[fn].y; o.z += old[fn].z; o.w += old[fn].w; }
      if (st32) *reinterpret_cast<float4*>(row + fn * 16) = o;
      if (imgt) {
        const uint2 w = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        *reinterpret_cast<uint2*>(imgt + ro + fn * 16) = w;
        if (!st32) ss += sq_bf16x2(w.x) + sq_bf16x2(w.y);
      }
      if (st32) ss += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
    }
  }
  if (p.sumsq)  // smem + 2 stages: 16 B the main loop never touches
    block_sum_store<4>(direct ? ss : 0.f, reinterpret_cast<float*>(smem + 2 * STAGE_BYTES), p.sumsq + blockIdx.x);
}

// dW tile t (+)= its pieces in order; grid (16, tiles), thread = 4 consecutive columns
__global__ __launch_bounds__(256) void reduce_bal_kernel(BalArgs p, int SA_actual, int SB_actual) {
  __shared__ float red[4];
  const int t = blockIdx.y;
  if (t < p.T_A && p.S_A == 1) return;  // accumulated in place by the GEMM
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int r = e >> 7, c = e & 127;
  int tr_, tc_;
  bal_tile_rc(t, p.tiles_r, p.tiles_c, p.group_rows, tr_, tc_);
  const size_t off = (size_t)(tr_ * 128 + r) * p.ldc + tc_ * 128 + c;
  float4 s = p.accumulate ? *reinterpret_cast<const float4*>(p.dW + off) : make_float4(0, 0, 0, 0);
  if (t < p.T_A) {
    for (int k = 0; k < SA_actual; ++k) {
      float4 v = *reinterpret_cast<const float4*>(p.wsA + (size_t)k * p.slab_stride + off);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else {
    const float* src = p.wsB + (size_t)(t - p.T_A) * (128 * 128) + e;
    const size_t zs = (size_t)p.T_B * (128 * 128);
    for (int k = 0; k < SB_actual; ++k) {
      float4 v = *reinterpret_cast<const float4*>(src + k * zs);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  float ss = 0.f;
  if (!p.img_only) {
    *reinterpret_cast<float4*>(p.dW + off) = s;
    ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
  }
  if (p.img) {
    const uint2 w = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
    *reinterpret_cast<uint2*>(p.img + off) = w;
    if (p.img_only) ss = sq_bf16x2(w.x) + sq_bf16x2(w.y);
  }
  if (p.sumsq_red) block_sum_store<4>(ss, red, p.sumsq_red + (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}


// ---- NT GEMM on 256 x 224 tiles, 8 waves, 8-phase schedule (dgrad launches with N = 896 under the two-stream backward) ------
// C[M][N] = A[M][K] B[N][K]^T (bf16 out, optional residual) with the tile geometry of gemm_tn_224_kernel and the operand
// staging of gemm_nt_256_kernel: half-tile images [128 rows][64 k] (16-B chunk swizzle, ds_read_b128 fragments),
//   H0 = A rows {first 32 of each wave row} | H1 = B rows {first 64 of each wave column}
//   H2 = B rows {last 48 of each wave column} (96 of 128 image rows used) | H3 = A rows {other 32 of each wave row}
// (operand rows are contraction-contiguous here, so any row selection is a whole 128-B line per row and K-tile).
// One block per tile, no K-splitting: 8192 x 896 outputs are 128 tiles - half the CUs. That is the point: in backward
// the wgrad stream keeps the other CUs busy, so a launch should minimise CU-time per flop (1.2-1.4 PFLOP/s-equivalent per
// occupied CU in the main loop against ~1.0 for the 128 x 128 kernel) rather than fill the chip by itself.
__global__ __launch_bounds__(512, 1) void gemm_nt_224_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HT = 128 * 128;  // half-tile bytes
  constexpr int KT = 4 * HT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 3, wc = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int nblk = p.tiles_r * p.tiles_c;
  int nid;
  {
    int id = blockIdx.x, xcd = id & 7, idx = id >> 3;
    int q = nblk >> 3, r = nblk & 7;
    nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tr_ = nid / p.tiles_c, tc_ = nid - tr_ * p.tiles_c;  // the column tiles of a row panel sit on one XCD
  const int row0 = tr_ * 256, col0 = tc_ * 224;
  const int nk = p.Kc / BK;
  const uint32_t lds0 = lds_addr(smem);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  uint32_t vo[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int P = i * 512 + tid, r = P >> 3, c = (P & 7) ^ lds_swz_key(r);
    const int ra0 = (r >> 5) * 64 + (r & 31), ra1 = ra0 + 32;
    const int rb0 = (r >> 6) * 112 + (r & 63);
    const int r1 = r < 96 ? r : 95;  // image rows 96..127 of H2 are unused: they repeat the last row
    const int rb1 = (r1 / 48) * 112 + 64 + (r1 % 48);
    vo[0][i] = (uint32_t)(((size_t)ra0 * p.lda + c * 8) * sizeof(bf16_t));
    vo[3][i] = (uint32_t)(((size_t)ra1 * p.lda + c * 8) * sizeof(bf16_t));
    vo[1][i] = (uint32_t)(((size_t)rb0 * p.ldb + c * 8) * sizeof(bf16_t));
    vo[2][i] = (uint32_t)(((size_t)rb1 * p.ldb + c * 8) * sizeof(bf16_t));
  }
  const bf16_t* Ab = p.A + (size_t)row0 * p.lda;
  const bf16_t* Bb = p.B + (size_t)col0 * p.ldb;
  auto issue_half = [&](int h, int t) {
    const bf16_t* base = ((h == 0 || h == 3) ? Ab : Bb) + (size_t)t * BK;
    const uint32_t dst = lds0 + (uint32_t)((t & 1) * KT + h * HT) + (uint32_t)wv * 1024u;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_sv(base, vo[h][i], __builtin_amdgcn_readfirstlane(dst + (uint32_t)(i * 8192)));
  };
  f32x4_t acc[4][7];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int ka = (l15 >> 1) & 7;
  auto frag = [&](const char* img, int Pb, int kk) -> uint4 {  // 16-row block Pb of a half-tile image, K-half kk
    return *reinterpret_cast<const uint4*>(img + (Pb * 16 + l15) * 128 + ((((g + 4 * kk) ^ ka ^ Pb) & 7) << 4));
  };
  uint4 afr[2][2], bg0[2][4], bg1[2][3];
  auto read_A = [&](const char* img) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 2; ++f) afr[kk][f] = frag(img, wr * 2 + f, kk);
  };
  auto read_B0 = [&](const char* img) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 4; ++f) bg0[kk][f] = frag(img, wc * 4 + f, kk);
  };
  auto read_B1 = [&](const char* img) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 3; ++f) bg1[kk][f] = frag(img, wc * 3 + f, kk);
  };
  auto mma0 = [&](int ah) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[ah * 2 + fm][fn] = mfma16(bg0[kk][fn], afr[kk][fm], acc[ah * 2 + fm][fn]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma1 = [&](int ah) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 3; ++fn) acc[ah * 2 + fm][4 + fn] = mfma16(bg1[kk][fn], afr[kk][fm], acc[ah * 2 + fm][4 + fn]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto ktile = [&](int t, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const char* buf = smem + (t & 1) * KT;
    if (!LAST) issue_half(0, t + 1);
    read_A(buf);
    read_B0(buf + HT);
    wait_ph<LAST>(0);
    raw_barrier();
    mma0(0);
    raw_barrier();
    if (!LAST) issue_half(1, t + 1);
    read_B1(buf + 2 * HT);
    wait_ph<LAST>(1);
    raw_barrier();
    mma1(0);
    raw_barrier();
    if (!LAST) issue_half(2, t + 1);
    read_A(buf + 3 * HT);
    raw_barrier();
    mma1(1);
    raw_barrier();
    if (!LAST) issue_half(3, t + 1);
    wait_ph<LAST>(2);
    raw_barrier();
    mma0(1);
    raw_barrier();
  };
  issue_half(0, 0);
  issue_half(1, 0);
  issue_half(2, 0);
  issue_half(3, 0);
  wait_vmcnt<4>();
  raw_barrier();
  if (wc == 1) raw_barrier();
  for (int t = 0; t + 1 < nk; ++t) ktile(t, std::false_type{});
  ktile(nk - 1, std::true_type{});
  if (wc == 0) raw_barrier();

  // epilogue: lane holds C[m][n .. n+3] per fragment (8-byte stores; optional residual)
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int m = row0 + wr * 64 + fm * 16 + l15;
    const size_t rowoff = (size_t)m * p.ldc;
    uint2 rr[7];
    if (p.resid) {
#pragma unroll
      for (int fn = 0; fn < 7; ++fn)
        rr[fn] = *reinterpret_cast<const uint2*>(p.resid + rowoff + col0 + wc * 112 + fn * 16 + g * 4);
    }
#pragma unroll
    for (int fn = 0; fn < 7; ++fn) {
      f32x4_t v = acc[fm][fn];
      if (p.resid) {
        v[0] += __uint_as_float(rr[fn].x << 16); v[1] += __uint_as_float(rr[fn].x & 0xffff0000u);
        v[2] += __uint_as_float(rr[fn].y << 16); v[3] += __uint_as_float(rr[fn].y & 0xffff0000u);
      }
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + rowoff + col0 + wc * 112 + fn * 16 + g * 4) = o;
    }
  }
}

// ---- wgrad on 256 x 224 tiles, 8 waves, 8-phase schedule ----------------------------------------------------------------
// The two big weight gradients of a layer (gate|up: [9728][896], down: [896][4864]; contraction over the M = 8192 tokens)
// hold 88 % of the wgrad flops and ran at ~930 TFLOP/s on the 128 x 128 one-barrier kernel. Every big matrix of the model
// has one dimension of 896 = 4 x 224 (and 1536-wide models one of 8960 = 40 x 224), so the phase-scheduled structure of
// gemm_nt_256_kernel is rebuilt here on 256 x 224 tiles: eight waves as 4 (256-side, 64 each) x 2 (224-side, 112 each),
// a wave owns 64 x 112 of the output (28 fragments, 112 accumulator VGPRs), 22 fragment reads per 56 MFMAs of a K-tile
// (128 x 128 kernel: 16 per 32). Both operands are stored contraction-major ([M][.]): they are DMA'd as stored into
// [64 kc][128 columns] half-tile images (256-B rows, 32-B block swizzle keyed by kc) and read with ds_read_b64_tr_b16,
// exactly the fragment addressing of gemm_tn_bal_kernel. A K-tile is four phases, one output quadrant of the wave each:
//   half-tiles  H0 = A columns 0..127 | H1 = B columns 0..127 | H2 = B columns 128..223 (96 of the 128 image columns
//               used) | H3 = A columns 128..255: every kc row of a half-tile is ONE contiguous 256-B (192-B) piece of the
//               stored operand. Wave row wr owns A columns {32 wr .. +31} of BOTH halves (its fragments 0,1 and 2,3), wave
//               column wc owns B columns {64 wc .. +63} of H1 and {128 + 48 wc .. +47} of H2.
//   phase 1: H0 + H1 -> (A0, B0) 16 MFMAs   2: H2 -> (A0, B1) 12   3: H3 -> (A1, B1) 12   4: H1 again (registers) -> (A1, B0) 16
// with the DMA of one half-tile of the next K-tile issued per phase, counted vmcnt(4), raised MFMA priority and the
// second wave column one barrier behind the first - the synchronisation skeleton of gemm_nt_256_kernel, unchanged.
// Work split: balanced K-splitting over the 256 CU slots (one block per CU): piece 0 of a tile accumulates straight into
// dW, later pieces go to fp32 slabs that reduce_224_kernel adds in piece order (no atomics, the same bits every run).
struct Tn224Args {
  const bf16_t* A;   // 256-side operand [M][lda]
  const bf16_t* B;   // 224-side operand [M][ldb]
  float* dW;         // element (i on the 256 side, j on the 224 side) at TR ? dW[j * ldw + i] : dW[i * ldw + j]
  float* slab;       // [slab index][256 * 224] fp32, tile-local in the orientation of dW
  int lda, ldb, ldw;
  int tiles_a, tiles_b, KS;
  int T_A, S_A, per_A, T_B, S_B, per_B, nA;
  int accumulate;
  bf16_t* img;  // nullable: bf16 image of dW (same indexing), written with every FINAL value of dW
  float* sumsq;      // nullable: GradSink slots of the GEMM kernel's blocks ...
  float* sumsq_red;  // ... and of reduce_224_kernel's (y-major)
  int img_only;      // final values go to img only (GradSink)
};
SLAM_DEVICE void tn224_tile(int t, int tiles_b, int& ta, int& tb) { ta = t / tiles_b; tb = t - ta * tiles_b; }

template <bool TR>
__global__ __launch_bounds__(512, 1) void gemm_tn_224_kernel(Tn224Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HT = 64 * 256;  // half-tile image: 64 kc rows x 256 B
  constexpr int KT = 4 * HT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 3, wc = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  // ---- which piece of which tile ----
  int t, ks0, ks1, z, pieces;
  size_t slab_idx = 0;
  {
    int b = blockIdx.x;
    pieces = b < p.nA ? p.S_A : p.S_B;
    if (b < p.nA) {
      z = b / p.T_A;
      const int idx = b - z * p.T_A;
      // block b runs on XCD b % 8: give each XCD a contiguous run of tiles
      const int xcd = idx & 7, i8 = idx >> 3, q = p.T_A >> 3, r = p.T_A & 7;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i8;
      ks0 = z * p.per_A;
      ks1 = min(p.KS, ks0 + p.per_A);
      if (z > 0) slab_idx = (size_t)(z - 1) * p.T_A + t;
    } else {
      b -= p.nA;
      z = b / p.T_B;
      const int idx = b - z * p.T_B;
      t = p.T_A + idx;
      ks0 = z * p.per_B;
      ks1 = min(p.KS, ks0 + p.per_B);
      if (z > 0) slab_idx = (size_t)(p.S_A - 1) * p.T_A + (size_t)(z - 1) * p.T_B + idx;
    }
  }
  int ta, tb;
  tn224_tile(t, p.tiles_b, ta, tb);
  const int a0 = ta * 256, b0 = tb * 224;
  const int nk = ks1 - ks0;
  const uint32_t lds0 = lds_addr(smem);
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // ---- DMA source offsets: 2 chunks of 16 B per lane per half-tile. LDS chunk P = i * 512 + tid sits in kc row P >> 4 at
  //      16-byte slot P & 15 and holds the source chunk (slot ^ (tr_key(kc) << 1)) of that row (32-B block swizzle) ----
  uint32_t vo[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int P = i * 512 + tid, kc = P >> 4, c = (P & 15) ^ (tr_key(kc) << 1);
    const int r = c * 8;  // first image column of the chunk
    const int colA0 = a0 + r, colA1 = a0 + 128 + r;
    const int colB0 = b0 + r;
    const int colB1 = b0 + 128 + (r < 96 ? r : 88);  // image columns 96..127 of H2 are unused: they repeat the last chunk
    vo[0][i] = (uint32_t)(((size_t)kc * p.lda + colA0) * sizeof(bf16_t));
    vo[3][i] = (uint32_t)(((size_t)kc * p.lda + colA1) * sizeof(bf16_t));
    vo[1][i] = (uint32_t)(((size_t)kc * p.ldb + colB0) * sizeof(bf16_t));
    vo[2][i] = (uint32_t)(((size_t)kc * p.ldb + colB1) * sizeof(bf16_t));
  }
  const bf16_t* Ak = p.A + (size_t)ks0 * BK * p.lda;
  const bf16_t* Bk = p.B + (size_t)ks0 * BK * p.ldb;
  auto issue_half = [&](int h, int t_) {
    const bf16_t* base = (h == 0 || h == 3) ? Ak + (size_t)t_ * BK * p.lda : Bk + (size_t)t_ * BK * p.ldb;
    const uint32_t dst = lds0 + (uint32_t)((t_ & 1) * KT + h * HT) + (uint32_t)wv * 1024u;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_sv(base, vo[h][i], __builtin_amdgcn_readfirstlane(dst + (uint32_t)(i * 8192)));
  };

  f32x4_t acc[4][7];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- fragments by transpose reads (lane (l15, g) of 16-column block Pb, K-half kk: kc = 32 kk + 8 g + (l15 >> 2) and + 4) ----
  const int trk = (l15 >> 2) | ((g & 1) << 2);
  const int tr_lane = (g * 8 + (l15 >> 2)) * 256 + (l15 & 3) * 8;
  auto frag = [&](const char* img, int Pb, int kk) -> uint4 {
    const char* q = img + tr_lane + kk * 32 * 256 + ((Pb ^ trk) << 5);
    const uint2 lo = lds_tr_read(q), hi = lds_tr_read(q + 4 * 256);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  uint4 afr[2][2], bg0[2][4], bg1[2][3];  // [kk][fragment]
  auto read_A = [&](const char* img) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 2; ++f) afr[kk][f] = frag(img, wr * 2 + f, kk);
  };
  auto read_B0 = [&](const char* img) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 4; ++f) bg0[kk][f] = frag(img, wc * 4 + f, kk);
  };
  auto read_B1 = [&](const char* img) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int f = 0; f < 3; ++f) bg1[kk][f] = frag(img, wc * 3 + f, kk);
  };
  // TR: lane holds 4 consecutive 256-side indices of one 224-side index (a-operand = A fragment); else the reverse
  auto mma0 = [&](int ah) {  // A half `ah` x B group 0 (fragments 0..3)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
          acc[ah * 2 + fm][fn] = TR ? mfma16(afr[kk][fm], bg0[kk][fn], acc[ah * 2 + fm][fn])
                                    : mfma16(bg0[kk][fn], afr[kk][fm], acc[ah * 2 + fm][fn]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma1 = [&](int ah) {  // A half `ah` x B group 1 (fragments 4..6)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 3; ++fn)
          acc[ah * 2 + fm][4 + fn] = TR ? mfma16(afr[kk][fm], bg1[kk][fn], acc[ah * 2 + fm][4 + fn])
                                        : mfma16(bg1[kk][fn], afr[kk][fm], acc[ah * 2 + fm][4 + fn]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto ktile = [&](int t_, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const char* buf = smem + (t_ & 1) * KT;
    // phase 1
    if (!LAST) issue_half(0, t_ + 1);
    read_A(buf);
    read_B0(buf + HT);
    wait_ph<LAST>(0);  // H2(t) landed -> read in phase 2
    raw_barrier();
    mma0(0);
    raw_barrier();
    // phase 2
    if (!LAST) issue_half(1, t_ + 1);
    read_B1(buf + 2 * HT);
    wait_ph<LAST>(1);  // H3(t) landed -> read in phase 3
    raw_barrier();
    mma1(0);
    raw_barrier();
    // phase 3
    if (!LAST) issue_half(2, t_ + 1);
    read_A(buf + 3 * HT);
    raw_barrier();
    mma1(1);
    raw_barrier();
    // phase 4
    if (!LAST) issue_half(3, t_ + 1);
    wait_ph<LAST>(2);  // H0(t+1), H1(t+1) landed -> read in phase 1 of the next K-tile
    raw_barrier();
    mma0(1);
    raw_barrier();
  };
  if (nk > 0) {
    issue_half(0, 0);
    issue_half(1, 0);
    issue_half(2, 0);
    issue_half(3, 0);
    wait_vmcnt<4>();
    raw_barrier();
    if (wc == 1) raw_barrier();  // second wave column: one barrier behind from here on
    for (int t_ = 0; t_ + 1 < nk; ++t_) ktile(t_, std::false_type{});
    ktile(nk - 1, std::true_type{});
    if (wc == 0) raw_barrier();  // balance the barrier count
  }

  // ---- epilogue: piece 0 adds into / stores dW, later pieces fill their slab ----
  const bool direct = z == 0;
  const bool add = direct && p.accumulate;
  float* base;
  int ld;
  if (direct) {
    base = TR ? p.dW + (size_t)b0 * p.ldw + a0 : p.dW + (size_t)a0 * p.ldw + b0;
    ld = p.ldw;
  } else {
    base = p.slab + slab_idx * (size_t)(256 * 224);
    ld = TR ? 256 : 224;
  }
  const bool fin = direct && pieces == 1;  // unsplit tile: these stores are the final values
  bf16_t* const imgt = (fin && p.img) ? p.img + (base - p.dW) : nullptr;
  const bool st32 = !(fin && p.img_only);  // a final value that is kept in bf16 only
  float ss = 0.f;
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    float4 old[7];
    float* ptr[7];
#pragma unroll
    for (int fn = 0; fn < 7; ++fn) {
      const int i = (fm >> 1) * 128 + wr * 32 + (fm & 1) * 16;
      const int j = fn < 4 ? wc * 64 + fn * 16 : 128 + wc * 48 + (fn - 4) * 16;
      ptr[fn] = TR ? base + (size_t)(j + l15) * ld + i + g * 4 : base + (size_t)(i + l15) * ld + j + g * 4;
      if (add) old[fn] = *reinterpret_cast<const float4*>(ptr[fn]);
    }
#pragma unroll
    for (int fn = 0; fn < 7; ++fn) {
      const f32x4_t v = acc[fm][fn];
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (add) { o.x += old[fn].x; o.y += old[fn].y; o.z += old[fn].z; o.w += old[fn].w; }
      if (st32) *reinterpret_cast<float4*>(ptr[fn]) = o;
      if (imgt) {
        const uint2 w = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        *reinterpret_cast<uint2*>(imgt + (ptr[fn] - base)) = w;
        if (!st32) ss += sq_bf16x2(w.x) + sq_bf16x2(w.y);
      }
      if (st32) ss += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
    }
  }
  if (p.sumsq)  // smem + KT * 2: 32 B behind the two K-tile buffers (the barrier counts of the wave columns are balanced here)
    block_sum_store<8>(fin ? ss : 0.f, reinterpret_cast<float*>(smem + 8 * 64 * 256), p.sumsq + blockIdx.x);
}

// dW tile t += its slabs (pieces 1..S-1) in piece order; grid (56, tiles): thread = 4 consecutive tile-local elements
template <bool TR>
__global__ __launch_bounds__(256) void reduce_224_kernel(Tn224Args p, int SA_act, int SB_act) {
  __shared__ float red[4];
  const int t = blockIdx.y;
  const bool inA = t < p.T_A;
  const int S = inA ? SA_act : SB_act;
  if (S <= 1) return;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;  // tile-local element, row-major in the slab orientation
  constexpr int LD = TR ? 256 : 224;
  const int r = e / LD, c = e - r * LD;
  int ta, tb;
  tn224_tile(t, p.tiles_b, ta, tb);
  float* dst = TR ? p.dW + (size_t)(tb * 224 + r) * p.ldw + ta * 256 + c : p.dW + (size_t)(ta * 256 + r) * p.ldw + tb * 224 + c;
  float4 s = *reinterpret_cast<const float4*>(dst);
  const size_t stride = (size_t)(inA ? p.T_A : p.T_B) * (256 * 224);
  const float* src = p.slab + (inA ? (size_t)t : (size_t)(p.S_A - 1) * p.T_A + (size_t)(t - p.T_A)) * (256 * 224) + e;
  for (int k = 0; k + 1 < S; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(src + k * stride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float ss = 0.f;
  if (!p.img_only) {
    *reinterpret_cast<float4*>(dst) = s;
    ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
  }
  if (p.img) {
    const uint2 w = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
    *reinterpret_cast<uint2*>(p.img + (dst - p.dW)) = w;
    if (p.img_only) ss = sq_bf16x2(w.x) + sq_bf16x2(w.y);
  }
  if (p.sumsq_red) block_sum_store<4>(ss, red, p.sumsq_red + (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s][i]   (fp32, deterministic split-K finish)
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n,
                                                            int splits, int accumulate, bf16_t* __restrict__ img,
                                                            int img_only, float* __restrict__ sumsq) {
  __shared__ float red[4];
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  float ss = 0.f;
  if (i < n) {
    float4 s = accumulate ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
    for (int k = 0; k < splits; ++k) {
      float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (!img_only) {
      *reinterpret_cast<float4*>(out + i) = s;
      ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
    }
    if (img) {
      const uint2 w = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
      *reinterpret_cast<uint2*>(img + i) = w;
      if (img_only) ss = sq_bf16x2(w.x) + sq_bf16x2(w.y);
    }
  }
  if (sumsq) block_sum_store<4>(ss, red, sumsq + blockIdx.x);
}

// nt_store  (field of GemmTune, kernels.h)
// group_rows: step-level A/B on MI355X (same box): 1 -> 32.4 ms, 2 -> 31.2, 3 -> 31.2, 4 -> 31.5, 8 -> 32.8  (field of GemmTune, kernels.h)

template <bool TA, bool TB, bool F32OUT, bool GLDS, bool PERM = false, bool MF32 = false>
int launch(GemmArgs a, int splits, hipStream_t st) {
  constexpr int lds = 2 * STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TA, TB, F32OUT, GLDS, PERM, MF32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  a.tiles_r = (a.R + BM - 1) / BM;
  a.group_rows = T().group_rows;
  a.nt_store = T().nt_store;
  dim3 grid(a.tiles_r * a.tiles_c, 1, splits);
  gemm_kernel<TA, TB, F32OUT, GLDS, PERM, MF32><<<grid, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}

}  // namespace

namespace slam {

// gemm_glds: 1 = LDS-DMA staging where the shape allows (default), 0 = register staging everywhere (parity tests)
// gemm_glds  (field of GemmTune, kernels.h)
// gemm_256: 1 = the 256 x 256 kernel when its fill criterion holds (default), 0 = never, 2 = whenever the shape allows (tests)
// gemm_256  (field of GemmTune, kernels.h)
// the 256 x 256 kernel runs one block per CU: worth it when the tiles fill most of whole rounds of the 256 CUs
// (gate|up forward: 1216 tiles = 4.75 rounds, 95 %; down-proj dgrad: 608 tiles = 2.4 rounds, 79 %: equal as a plain
// GEMM, +1 % step throughput with its fused SwiGLU-backward epilogue; N = 1536 at M = 16384: 384 tiles = 1.5 rounds,
// 75 %, still +4.7 % on the Qwen2.5-1.5B-shaped step because its contractions are long)
static bool use_256(const GemmArgs& a) {
  if (!T().g256 || (a.R % 256) || (a.Cn % 256) || (a.Kc % BK) || a.Kc < 2 * BK) return false;
  const int tiles = (a.R / 256) * (a.Cn / 256);
  if (T().g256 == 2) return tiles >= 256;  // forced (tests / A-B)
  return tiles >= 256 && (double)tiles / (double)(((tiles + 255) / 256) * 256) >= 0.74;
}
// the down-proj dgrad with the fused SwiGLU backward on the 256 x 256 kernel (1) or on the 128 x 128 kernel (0: two blocks
// per CU, so that one block's HBM-bound epilogue sits beside another block's - or a concurrent wgrad's - MFMA loop)
// gemm_256_dswiglu  (field of GemmTune, kernels.h)
// group_rows_256: 256-row tile groups: 1 -> 145 us, 2 -> 135, 4 -> 133, 8 -> 133 (gate|up forward, plain)  (field of GemmTune, kernels.h)
// persistent blocks (one per CU) walking the tile list with the DMA stream running across tile boundaries: 0 = one
// block per tile, 1 = whenever there are more tiles than CUs and the epilogue has no bias / residual / RoPE operand.
// Same box, interleaved: gate|up forward + SwiGLU 147.4 -> 136.3 us, plain 130.0 -> 121.5, down-proj dgrad + dSwiGLU
// 109.1 -> 104.7, LM head of the 152k vocabulary 6294 -> 6106; Slam-358M step 313.4 k -> 319.7 k tok/s (two pairs)
// gemm_256_persist  (field of GemmTune, kernels.h)
static int launch_256(GemmArgs a, hipStream_t st) {
  static int cus_dev = 0;
  if (!cus_dev) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
    if (e != hipSuccess) return (int)e;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    cus_dev = n & ~7;
  }
  a.tiles_r = a.R / 256;
  a.tiles_c = a.Cn / 256;
  a.group_rows = T().group_rows_256;
  a.group_cols = T().group_cols_256;
  a.nt_store = T().nt_store;
  a.stagger_ticks = a.gu ? T().g256_stagger_dswiglu : T().g256_stagger;
  if (a.stagger_ticks > (a.Kc / BK) * 100) a.stagger_ticks = (a.Kc / BK) * 100;  // never more than ~a K loop (1 us per K-tile)
  a.cohorts = T().g256_cohorts;
  a.batch_epilogue_loads = T().g256_batch_loads;
  const int tiles = a.tiles_r * a.tiles_c;
  // gemm_256_persist_cus > 0: the persistent grid leaves CUs free (a multiple of 8 blocks: one slot count per XCD) - under data
  // parallelism RCCL's kernels need somewhere to start while 256 one-per-CU blocks hold every CU (bench.py extras.dp_variants)
  const int cus_all = cus_dev;
  const int want = T().g256_persist_cus > 0 ? (T().g256_persist_cus & ~7) : cus_all;
  const int cus = want >= 8 && want < cus_all ? want : cus_all;
  const bool persist = T().g256_persist && tiles > cus && !a.bias && !a.resid && !a.rope_cos;
  if (T().mf32 && T().g256_w4 && persist) {  // four waves of 128 x 128, persistent
    static bool attr4 = false;
    if (!attr4) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_w4_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
      if (e != hipSuccess) return (int)e;
      attr4 = true;
    }
    gemm_nt_w4_kernel<8><<<cus, 256, 8 * 128 * 128, st>>>(a);
    return (int)hipGetLastError();
  }
  if (T().mf32) {
    if (persist) gemm_nt_256_kernel<true, true><<<cus, 512, 8 * 128 * 128, st>>>(a);
    else gemm_nt_256_kernel<false, true><<<tiles, 512, 8 * 128 * 128, st>>>(a);
  } else {
    if (persist && T().g256_roles) gemm_nt_256_kernel<true, false, true><<<cus, 512, 8 * 128 * 128, st>>>(a);
    else if (persist) gemm_nt_256_kernel<true, false><<<cus, 512, 8 * 128 * 128, st>>>(a);
    else gemm_nt_256_kernel<false, false><<<tiles, 512, 8 * 128 * 128, st>>>(a);
  }
  return (int)hipGetLastError();
}

// 256 x 224 NT kernel: only in "shared" mode (the engine's two-stream backward), plain or residual epilogue
// shared: set by the engine around slam_backward when the wgrad stream is on  (field of GemmTune, kernels.h)
// nt224: 0 = off, 1 = in shared mode (default), 2 = whenever the shape allows (tests)  (field of GemmTune, kernels.h)
// nt224_min_k: long contractions only (gate|up dgrad, K = 9728): interleaved A/B x3 on the Slam-358M step 311.5-312.0k vs 310.3-310.6k tok/s; with the K = 896 / 1152 dgrads as well: no gain  (field of GemmTune, kernels.h)
static bool use_nt224(const GemmArgs& a) {
  if (!T().nt224 || (a.R % 256) || (a.Cn % 224) || (a.Kc % BK) || a.Kc < 2 * BK) return false;
  if (a.bias || a.act || a.gu || a.rope_cos) return false;
  if (T().nt224 == 2) return true;
  return T().shared && a.Kc >= T().nt224_min_k && (a.R / 256) * (a.Cn / 224) >= 64;
}
static int launch_nt224(GemmArgs a, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_224_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  a.tiles_r = a.R / 256;
  a.tiles_c = a.Cn / 224;
  gemm_nt_224_kernel<<<a.tiles_r * a.tiles_c, 512, 8 * 128 * 128, st>>>(a);
  return (int)hipGetLastError();
}
// the 128 x 128 NT DMA kernel on either MFMA shape
static int launch_nt_128(const GemmArgs& a, hipStream_t st) {
  return T().mf32 ? launch<false, false, false, true, true, true>(a, 1, st) : launch<false, false, false, true, true>(a, 1, st);
}
// NT launches whose rows / contraction are whole tiles: 256 x 256 8-phase kernel or the 128 x 128 DMA kernel
static int launch_nt_dma(const GemmArgs& a, hipStream_t st) {
  if (use_nt224(a)) return launch_nt224(a, st);
  if (use_256(a)) return launch_256(a, st);
  return launch_nt_128(a, st);
}
// gemm_tn_dma  (field of GemmTune, kernels.h)

static int check_dims(int R, int Cn, int Kc, int lda, int ldb, int ldc) {
  if (R <= 0 || Cn <= 0 || Kc <= 0) return -1;
  if ((Cn & 7) || (lda & 7) || (ldb & 7) || (ldc & 3)) return -1;
  return 0;
}

// Y[M,N] = X[M,K] W[N,K]^T (+bias[N]) (+resid[M,N]); bf16 in/out, fp32 accumulate.
int gemm_nt(const bf16_t* X, const bf16_t* W, bf16_t* Y, const bf16_t* bias, const bf16_t* resid, int M,
            int N, int K, hipStream_t st) {
  if (check_dims(M, N, K, K, K, N) || (K & 7)) return -1;
  GemmArgs a{X, W, Y, bias, resid, nullptr, nullptr, M, N, K, K, K, N, ((K + BK - 1) / BK) * BK, (M + BM - 1) / BM,
             (N + BN - 1) / BN};
  const bool dma_ok = (K % BK == 0) && (N % BN == 0);
  if (dma_ok && T().glds) return launch_nt_dma(a, st);
  return launch<false, false, false, false>(a, 1, st);
}

// QKV projection with bias and rotate-half RoPE applied to the first rope_heads heads in the epilogue
int gemm_nt_rope(const bf16_t* X, const bf16_t* W, bf16_t* Y, const bf16_t* bias, const float* cs, const float* sn,
                 const float* csq, const float* snq, int q_heads, int rope_heads, int M, int N, int K, hipStream_t st) {
  if (check_dims(M, N, K, K, K, N) || (K % BK) || (N % BN)) return -1;
  GemmArgs a{X, W, Y, bias, nullptr, nullptr, nullptr, M, N, K, K, K, N, K, (M + BM - 1) / BM, N / BN, cs, sn, rope_heads};
  a.rope_cos_q = csq ? csq : cs; a.rope_sin_q = snq ? snq : sn; a.rope_q_heads = csq ? q_heads : 0;
  return launch_nt_dma(a, st);
}

int gemm_nt_swiglu(const bf16_t* X, const bf16_t* W, bf16_t* Y, bf16_t* act, int M, int N, int K, hipStream_t st) {
  if (check_dims(M, N, K, K, K, N) || (K % BK) || (N % BN)) return -1;
  GemmArgs a{X, W, Y, nullptr, nullptr, act, nullptr, M, N, K, K, K, N, K, (M + BM - 1) / BM, N / BN};
  return launch_nt_dma(a, st);
}

// d(act)[M,N] = dY[M,K] Wt[N,K]^T is never stored: gu [M,2N] (32-column gate/up blocks) is rewritten in
// place with d(gate|up) = SwiGLU'(gate, up) * d(act).
int gemm_nt_dswiglu(const bf16_t* dY, const bf16_t* Wt, bf16_t* gu, int M, int N, int K, hipStream_t st) {
  if (check_dims(M, N, K, K, K, N) || (K % BK) || (N % BN) || (N % 32)) return -1;
  GemmArgs a{dY, Wt, nullptr, nullptr, nullptr, nullptr, gu, M, N, K, K, K, N, K, (M + BM - 1) / BM, N / BN};
  if (!T().g256_dswiglu) return launch_nt_128(a, st);
  return launch_nt_dma(a, st);
}

// dX[M,K] = dY[M,N] W[N,K] (+resid[M,K]); contraction over N.
int gemm_nn(const bf16_t* dY, const bf16_t* W, bf16_t* dX, const bf16_t* resid, int M, int N, int K,
            hipStream_t st) {
  if (check_dims(M, K, N, N, K, K) || (N & 7)) return -1;
  GemmArgs a{dY, W, dX, nullptr, resid, nullptr, nullptr, M, K, N, N, K, K, ((N + BK - 1) / BK) * BK, (M + BM - 1) / BM,
             (K + BN - 1) / BN};
  return launch<false, true, false, false>(a, 1, st);
}

// tn_splits_override  (field of GemmTune, kernels.h)
int gemm_tn_splits(int M, int N, int K) {
  if (T().tn_splits_override > 0) return T().tn_splits_override;
  int tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN);
  // measured on MI355X (tools/gemm_bench.py split sweep): ~380 blocks for the few-tile weights
  // (wqkv 63 tiles -> 6, wo 49 -> 8), ~800 for the big ones (wd 266 -> 3, wgu 532 -> 2)
  int s = tiles < 128 ? (380 + tiles / 2) / tiles : (800 + tiles / 2) / tiles;
  int maxs = (M + 511) / 512;  // at least 8 k-steps per slice
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}
// balanced plan for the DMA-eligible shapes (see gemm_tn_bal_kernel)
// tn_balanced  (field of GemmTune, kernels.h)
static const int BAL_SLOTS = 512;  // 2 blocks per CU on the 256-CU MI355X
struct BalPlan { int T_A, S_A, per_A, SA_act, T_B, S_B, per_B, SB_act; };
static bool tn_bal_ok(int M, int N, int K) {
  return T().tn_balanced && T().tn_dma == 1 && (N % BM == 0) && (K % BN == 0) && (M % BK == 0);
}
// bal_bg_max_split: background launches of the 128 x 128 balanced kernel: limit on pieces per tile (measured on the Slam-358M step: 1 -> 291.7k, 2 -> 306.4k, 3 -> 312.2k, 4 -> 312.4k, 5 -> 311.3k, 6 -> 310.6k, 8 -> 308.7k tok/s)  (field of GemmTune, kernels.h)
static BalPlan tn_bal_plan(int M, int N, int K, int max_split = 8) {
  const int T = (N / BM) * (K / BN), KS = M / BK;
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  const int OVH = 3;  // K-steps' worth of prologue + epilogue per piece (estimate used to choose between plans)
  BalPlan best{};
  int best_cost = 1 << 30;
  auto consider = [&](int S, int T_A) {
    if (S < 1 || S > max_split || S > KS || T_A < 0 || T_A > T) return;
    BalPlan pl{};
    pl.T_A = T_A; pl.S_A = S; pl.per_A = cdiv(KS, S); pl.SA_act = T_A ? cdiv(KS, pl.per_A) : 0;
    pl.T_B = T - T_A;
    int cost = T_A ? cdiv(T_A * pl.SA_act, BAL_SLOTS) * (pl.per_A + OVH) : 0;
    if (pl.T_B) {
      int sb = BAL_SLOTS / pl.T_B;
      if (sb > 16) sb = 16;
      if (sb > 2 * max_split) sb = 2 * max_split;
      if (sb < 1) sb = 1;
      if (sb > KS) sb = KS;
      pl.S_B = sb; pl.per_B = cdiv(KS, sb); pl.SB_act = cdiv(KS, pl.per_B);
      cost += cdiv(pl.T_B * pl.SB_act, BAL_SLOTS) * (pl.per_B + OVH);
    }
    // slab traffic (written once, read once) in K-step units: ~0.5 K-steps per MB at ~5 TB/s
    const double slab_mb = ((pl.S_A > 1 ? (double)pl.SA_act * pl.T_A : 0.0) + (double)pl.SB_act * pl.T_B) * (128 * 128 * 4) / 1e6;
    cost += (int)(0.5 * slab_mb);
    if (cost < best_cost) { best_cost = cost; best = pl; }
  };
  for (int S = 1; S <= max_split; ++S) {
    consider(S, T);                                                  // everything in equal pieces
    int full = (T * S / BAL_SLOTS) * BAL_SLOTS / S;                  // tiles that make whole rounds
    if (S > 1) full &= ~7;
    if (full > 0 && full < T && max_split > 1) consider(S, full);
  }
  return best;
}

// ---- 256 x 224 phase-scheduled wgrad: shape test, balanced plan, launch -------------------------------------------------
// tn224: 0 = off, 1 = when the plan fills the chip (default), 2 = whenever the shape allows (tests)
// tn224  (field of GemmTune, kernels.h)
// tn224_min_m, tn224_max_split  (field of GemmTune, kernels.h)
// "background" launches (the engine's wgrad side stream: other kernels fill whatever CUs a launch leaves free, so what
// counts is CU-time per flop, not chip fill): no K-splitting at all - one block per tile walks the whole contraction (no
// slabs, no reduce pass, 128 K-steps of main loop per 229 KB epilogue), from much shorter contractions on.
// Measured on the Slam-358M step (same box, twice): 304.6k tok/s vs 292.5k with the balanced 128 x 128 kernel on that
// stream (+4.1 %); limits of 2 / 3 / 16 pieces: 299.0k / 288.1k / 289.4k.
// tn224_bg_min_m, tn224_bg_max_split  (field of GemmTune, kernels.h)
struct Plan224 { int T_A, S_A, per_A, T_B, S_B, per_B, slabs; bool tr; int tiles_a, tiles_b; };
// orientation: 0 = none, 1 = dW[n][k] with n on the 256 side (N % 256 == 0, K % 224 == 0), 2 = transposed store (k on the 256 side)
static int tn224_orient(int M, int N, int K, int background = 0) {
  if (!T().tn224 || (M % BK)) return 0;
  const bool o1 = (N % 256 == 0) && (K % 224 == 0), o2 = (K % 256 == 0) && (N % 224 == 0);
  if (!o1 && !o2) return 0;
  const long tiles = (long)N * K / (256 * 224);
  // Measured (MI355X, round 2): the main loop runs 1.24 PFLOP/s against 0.96 for the 128 x 128 kernel on a 65,536-long
  // contraction, but at M = 8192 every piece is 13..64 K-steps long and the one-block-per-CU epilogue (229 KB of fp32 per
  // piece, nothing to overlap it with) plus the slab pass give the gain back: gate|up weight 148-157 us vs 149-153,
  // down weight 77 vs 86 us, Slam-358M step 291.1k vs 294.3k tok/s. Default: contractions of 16,384 tokens and more.
  if (T().tn224 != 2 && (background ? (tiles < 32 || M < T().tn224_bg_min_m) : (tiles < 64 || M < T().tn224_min_m))) return 0;
  return o1 ? 1 : 2;
}
static Plan224 tn224_plan(int M, int N, int K, int orient, int max_split = -1) {
  Plan224 best{};
  if (max_split < 1) max_split = T().tn224_max_split;
  const int SLOTS = 256;  // one block per CU
  const int T = (N / (orient == 1 ? 256 : 224)) * (K / (orient == 1 ? 224 : 256)), KS = M / BK;
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  const int OVH = 3;  // K-steps' worth of prologue + epilogue per piece
  int best_cost = 1 << 30;
  auto consider = [&](int S, int T_A) {
    if (S < 1 || S > max_split || S > KS || T_A < 0 || T_A > T) return;
    Plan224 pl{};
    pl.T_A = T_A; pl.per_A = cdiv(KS, S); pl.S_A = T_A ? cdiv(KS, pl.per_A) : 1;
    pl.T_B = T - T_A; pl.S_B = 1; pl.per_B = KS;
    int cost = T_A ? cdiv(T_A * pl.S_A, SLOTS) * (pl.per_A + OVH) : 0;
    if (pl.T_B) {
      int sb = SLOTS / pl.T_B;
      if (sb > max_split) sb = max_split;
      if (sb < 1) sb = 1;
      if (sb > KS) sb = KS;
      pl.per_B = cdiv(KS, sb); pl.S_B = cdiv(KS, pl.per_B);
      cost += cdiv(pl.T_B * pl.S_B, SLOTS) * (pl.per_B + OVH);
    }
    pl.slabs = (pl.S_A - 1) * pl.T_A + (pl.S_B - 1) * pl.T_B;
    // slab traffic (written in the kernel, read by the reduce): ~0.12 K-steps of this kernel per MB
    cost += (int)(0.12 * pl.slabs * (256 * 224 * 4) / 1e6);
    if (cost < best_cost) { best_cost = cost; best = pl; }
  };
  for (int S = 1; S <= max_split; ++S) {
    consider(S, T);
    const int full = (T * S / SLOTS) * SLOTS / S;
    if (full > 0 && full < T) consider(S, full);
  }
  best.tr = orient == 2;
  best.tiles_a = orient == 1 ? N / 256 : K / 256;
  best.tiles_b = orient == 1 ? K / 224 : N / 224;
  return best;
}
static size_t tn224_workspace_bytes(int Mmax, int N, int K) {
  // sized for every plan that can run later: any contraction length up to Mmax (packed micro-batches vary) under every
  // split limit the options can take (forced on, foreground or background): any eligible shape reserves its slabs
  const bool o1 = (N % 256 == 0) && (K % 224 == 0), o2 = (K % 256 == 0) && (N % 224 == 0);
  if (!o1 && !o2) return 0;
  int slabs = 0;
  for (int ks = 1; ks <= Mmax / BK; ++ks)
    for (int ms = 1; ms <= 16; ++ms) {
      const Plan224 pl = tn224_plan(ks * BK, N, K, o1 ? 1 : 2, ms);
      if (pl.slabs > slabs) slabs = pl.slabs;
    }
  return (size_t)slabs * 256 * 224 * sizeof(float);
}
static int launch_tn224(const bf16_t* dY, const bf16_t* X, float* dW, int accumulate, int M, int N, int K, int ldy, int ldx,
                        float* ws, int orient, int background, hipStream_t st, bf16_t* img, GradSink* sink) {
  constexpr int LDS = 8 * 64 * 256 + 64;  // two K-tile buffers + the block-sum scratch of the GradSink epilogue
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_224_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_224_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const Plan224 pl = tn224_plan(M, N, K, orient, background ? T().tn224_bg_max_split : T().tn224_max_split);
  Tn224Args a{};
  a.A = orient == 1 ? dY : X; a.lda = orient == 1 ? ldy : ldx;
  a.B = orient == 1 ? X : dY; a.ldb = orient == 1 ? ldx : ldy;
  a.dW = dW; a.ldw = K; a.slab = ws;
  a.tiles_a = pl.tiles_a; a.tiles_b = pl.tiles_b; a.KS = M / BK;
  a.T_A = pl.T_A; a.S_A = pl.S_A; a.per_A = pl.per_A; a.T_B = pl.T_B; a.S_B = pl.S_B; a.per_B = pl.per_B;
  a.nA = pl.T_A * pl.S_A;
  a.accumulate = accumulate;
  a.img = img;
  const int nblk = a.nA + pl.T_B * pl.S_B;
  const int T = pl.T_A + pl.T_B;
  if (sink) {
    sink->used = nblk + (pl.slabs ? 56 * T : 0);
    if (sink->img_only && !img) return -1;
    if (sink->sumsq && sink->used > sink->cap) return -3;
    a.img_only = sink->img_only;
    a.sumsq = sink->sumsq;
    a.sumsq_red = sink->sumsq ? sink->sumsq + nblk : nullptr;
  }
  if (pl.tr) {
    gemm_tn_224_kernel<true><<<nblk, 512, LDS, st>>>(a);
    if (pl.slabs) reduce_224_kernel<true><<<dim3(56, T), 256, 0, st>>>(a, pl.S_A, pl.S_B);
  } else {
    gemm_tn_224_kernel<false><<<nblk, 512, LDS, st>>>(a);
    if (pl.slabs) reduce_224_kernel<false><<<dim3(56, T), 256, 0, st>>>(a, pl.S_A, pl.S_B);
  }
  return (int)hipGetLastError();
}

static size_t bal_plan_bytes(const BalPlan& pl, int N, int K) {
  return ((pl.S_A > 1 ? (size_t)pl.SA_act * N * K : 0) + (size_t)pl.SB_act * pl.T_B * 128 * 128) * sizeof(float);
}
// Workspace for dW[N,K] with contractions of up to Mmax rows: the maximum over EVERY plan gemm_tn can choose later - any
// contraction length M <= Mmax (the planners re-plan from the runtime M, which varies per packed micro-batch) under any
// split limit (foreground 8, the side stream's "gemm_tn_bal_bg_max_split") - not just the plan of M = Mmax (round 2 sized
// for that one: with H = 256, I = 1024, Mmax = 2048 the plan at M = 1600 needed 10 MiB of an 8 MiB allocation). gemm_tn
// also checks the chosen plan against the capacity it is given and fails instead of writing past it.
size_t gemm_tn_workspace_bytes(int Mmax, int N, int K) {
  size_t need = (size_t)gemm_tn_splits(Mmax, N, K) * N * K * sizeof(float);  // split-K fallback: grows with M
  if ((N % BM == 0) && (K % BN == 0))
    for (int ks = 1; ks <= Mmax / BK; ++ks)
      for (int ms = 1; ms <= 8; ++ms) {
        const size_t b = bal_plan_bytes(tn_bal_plan(ks * BK, N, K, ms), N, K);
        if (b > need) need = b;
      }
  const size_t t224 = tn224_workspace_bytes(Mmax, N, K);
  if (t224 > need) need = t224;
  return need;
}

// dW[N,K] (fp32) (+)= dY[M,N]^T X[M,K]; contraction over M; split-K partials in `ws`.
size_t gemm_tn_sumsq_slots(int N, int K) {
  // GEMM blocks <= 16 pieces per 128 x 128 tile (the 256 x 224 tiles are larger), reduce blocks = one per 1024 elements
  const size_t t128 = (size_t)((N + BM - 1) / BM) * ((K + BN - 1) / BN);
  return 32 * t128 + 64;
}

int gemm_tn(const bf16_t* dY, const bf16_t* X, float* dW, int accumulate, int M, int N, int K, int ldy,
            int ldx, float* ws, size_t ws_bytes, hipStream_t st, int background, bf16_t* img, GradSink* sink) {
  if (check_dims(N, K, M, ldy, ldx, K) || (N & 7)) return -1;
  if (sink) {
    sink->used = 0;
    if (sink->img_only && !img) return -1;
  }
  if (const int orient = tn224_orient(M, N, K, background)) {
    const Plan224 pl = tn224_plan(M, N, K, orient, background ? T().tn224_bg_max_split : T().tn224_max_split);
    if ((size_t)pl.slabs * 256 * 224 * sizeof(float) > ws_bytes) return -3;
    return launch_tn224(dY, X, dW, accumulate, M, N, K, ldy, ldx, ws, orient, background, st, img, sink);
  }
  if (tn_bal_ok(M, N, K)) {
    constexpr int LDS = 2 * STAGE_BYTES + 64;  // + the block-sum scratch of the GradSink epilogue
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bal_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      attr = true;
    }
    const BalPlan pl = tn_bal_plan(M, N, K, background ? T().bal_bg_max_split : 8);
    if (bal_plan_bytes(pl, N, K) > ws_bytes) return -3;
    BalArgs a{};
    a.A = dY; a.B = X; a.dW = dW; a.lda = ldy; a.ldb = ldx; a.ldc = K;
    a.tiles_r = N / BM; a.tiles_c = K / BN; a.KS = M / BK; a.group_rows = T().group_rows;
    a.T_A = pl.T_A; a.S_A = pl.S_A; a.per_A = pl.per_A; a.T_B = pl.T_B; a.S_B = pl.S_B; a.per_B = pl.per_B;
    a.nA = pl.T_A * pl.SA_act;
    a.accumulate = accumulate;
    a.slab_stride = (size_t)N * K;
    a.img = img;
    a.wsA = ws;
    a.wsB = ws + (pl.S_A > 1 ? (size_t)pl.SA_act * N * K : 0);
    const int nblk = a.nA + pl.T_B * pl.SB_act;
    const bool red = pl.S_A > 1 || pl.T_B > 0;
    if (sink) {
      sink->used = nblk + (red ? 16 * a.tiles_r * a.tiles_c : 0);
      if (sink->sumsq && sink->used > sink->cap) return -3;
      a.img_only = sink->img_only;
      a.sumsq = sink->sumsq;
      a.sumsq_red = sink->sumsq ? sink->sumsq + nblk : nullptr;
    }
    gemm_tn_bal_kernel<<<nblk, 256, LDS, st>>>(a);
    if (red) reduce_bal_kernel<<<dim3(16, a.tiles_r * a.tiles_c), 256, 0, st>>>(a, pl.SA_act, pl.SB_act);
    return (int)hipGetLastError();
  }
  int splits = gemm_tn_splits(M, N, K);
  int per = (((M + splits - 1) / splits) + BK - 1) / BK * BK;
  splits = (M + per - 1) / per;
  if ((size_t)splits * N * K * sizeof(float) > ws_bytes) return -3;
  GemmArgs a{dY, X, ws, nullptr, nullptr, nullptr, nullptr, N, K, M, ldy, ldx, K, per, (N + BM - 1) / BM, (K + BN - 1) / BN};
  const bool dma_ok = T().tn_dma && (N % BM == 0) && (K % BN == 0) && (M % BK == 0);
  int e = dma_ok ? launch<true, true, true, true>(a, splits, st) : launch<true, true, true, false>(a, splits, st);
  if (e) return e;
  size_t n = (size_t)N * K;
  const unsigned rblk = (unsigned)((n / 4 + 255) / 256);
  if (sink) {
    sink->used = (int)rblk;
    if (sink->sumsq && sink->used > sink->cap) return -3;
  }
  reduce_splits_kernel<<<rblk, 256, 0, st>>>(ws, dW, n, splits, accumulate, img, sink ? sink->img_only : 0, sink ? sink->sumsq : nullptr);
  return (int)hipGetLastError();
}

}  // namespace slam
