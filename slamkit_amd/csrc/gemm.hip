// bf16 MFMA GEMM for gfx950: C[R,Cn] = A·Bᵀ with per-operand "contraction-major" staging.
//
// Replaces the torch Linear kernels behind Qwen2Attention.{q,k,v,o}_proj and
// Qwen2MLP.{gate,up,down}_proj (site-packages transformers/models/qwen2/modeling_qwen2.py:41-48,
// 189-192) and their autograd dgrad / wgrad; SURVEY.md §8a rows T3, T6, T7, T8.
//
//   forward : Y[M,N]  = X[M,K]  · W[N,K]ᵀ      (TA=0, TB=0)   both operands contraction-contiguous
//   dgrad   : dX[M,K] = dY[M,N] · W[N,K]       (TA=0, TB=1)   B stored [contraction][cols]
//   wgrad   : dW[N,K] = dYᵀ[N,M] · X[M,K]      (TA=1, TB=1)   A stored [contraction][rows]
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 fragments.
// Both operand tiles live in LDS as [128][64] bf16 with a 16-byte-chunk XOR swizzle (common.h), so
// the MFMA inner loop is identical for all three forms; only global->LDS staging differs
// (direct 16-byte copies, LDS-DMA, or an 8x8 in-register transpose).
// MFMA roles are swapped (a-operand = column tile, b-operand = row tile) so a lane ends up holding
// four consecutive output columns of one row -> 8-byte bf16 / 16-byte fp32 stores.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;  // one operand tile: 128 rows x 128 B

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const bf16_t* bias;
  const bf16_t* resid;
  int R, Cn, Kc;
  int lda, ldb, ldc;
  int kc_per_split;
  int tiles_r, tiles_c;
};

SLAM_DEVICE uint32_t comp4(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ---- direct staging: operand stored [rows][contraction], 4 x 16 B per thread ----------------
SLAM_DEVICE void load_direct(const bf16_t* G, int ld, int nrows, int row0, int k0, int kend, int tid,
                             uint4* r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + 256 * i;
    int row = q >> 3, c = q & 7;
    int gr = row0 + row, gk = k0 + c * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < nrows && gk < kend) v = *reinterpret_cast<const uint4*>(G + (size_t)gr * ld + gk);
    r[i] = v;
  }
}
SLAM_DEVICE void store_direct(char* tile, int tid, const uint4* r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + 256 * i;
    int row = q >> 3, c = q & 7;
    *reinterpret_cast<uint4*>(tile + lds_tile_off(row, c)) = r[i];
  }
}

// ---- transposed staging: operand stored [contraction][rows]; one 8(kc) x 8(rows) unit per thread,
//      128 units per tile (unit u: rows (u&15)*8.., kc (u>>4)*8..) -------------------------------
SLAM_DEVICE void load_transposed(const bf16_t* G, int ld, int nrows, int row0, int k0, int kend, int u,
                                 uint4* r) {
  int rb = u & 15, kb = u >> 4;
  int gr = row0 + rb * 8;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    int gk = k0 + kb * 8 + kk;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gk < kend && gr < nrows) v = *reinterpret_cast<const uint4*>(G + (size_t)gk * ld + gr);
    r[kk] = v;
  }
}
SLAM_DEVICE void store_transposed(char* tile, int u, const uint4* r) {
  int rb = u & 15, kb = u >> 4;
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

// ---- LDS-DMA staging (direct operands only): the LDS image is lane-linear, so the swizzle is
//      applied to the per-lane SOURCE chunk; rows past the end are clamped (their products only
//      reach output rows that are never stored). ------------------------------------------------
SLAM_DEVICE void glds_tile(const bf16_t* G, int ld, int nrows, int row0, int k0, int tid, char* tile) {
  int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int P = i * 256 + tid;
    int row = P >> 3, cs = P & 7;
    int c = cs ^ lds_swz_key(row);
    int gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    const bf16_t* src = G + (size_t)gr * ld + k0 + c * 8;
    char* dst = tile + (i * 256 + wave * 64) * 16;  // wave-uniform; hardware adds lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

template <bool TA, bool TB, bool F32OUT, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [stage][A|B]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
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
  const int row0 = (nid / p.tiles_c) * BM;
  const int col0 = (nid % p.tiles_c) * BN;
  const int kbeg = blockIdx.z * p.kc_per_split;
  const int kend = min(p.Kc, kbeg + p.kc_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  uint4 sa[8], sb[8];

  auto stage_load = [&](int t) {
    const int k0 = kbeg + t * BK;
    if constexpr (TA && TB) {
      if (tid < 128) load_transposed(p.A, p.lda, p.R, row0, k0, kend, tid, sa);
      else load_transposed(p.B, p.ldb, p.Cn, col0, k0, kend, tid - 128, sa);
    } else {
      if constexpr (TA) { if (tid < 128) load_transposed(p.A, p.lda, p.R, row0, k0, kend, tid, sa); }
      else load_direct(p.A, p.lda, p.R, row0, k0, kend, tid, sa);
      if constexpr (TB) { if (tid < 128) load_transposed(p.B, p.ldb, p.Cn, col0, k0, kend, tid, sb); }
      else load_direct(p.B, p.ldb, p.Cn, col0, k0, kend, tid, sb);
    }
  };
  auto stage_store = [&](int s) {
    char* At = smem + s * 2 * TILE_BYTES;
    char* Bt = At + TILE_BYTES;
    if constexpr (TA && TB) {
      if (tid < 128) store_transposed(At, tid, sa);
      else store_transposed(Bt, tid - 128, sa);
    } else {
      if constexpr (TA) { if (tid < 128) store_transposed(At, tid, sa); }
      else store_direct(At, tid, sa);
      if constexpr (TB) { if (tid < 128) store_transposed(Bt, tid, sb); }
      else store_direct(Bt, tid, sb);
    }
  };
  auto stage_glds = [&](int t, int s) {
    const int k0 = kbeg + t * BK;
    char* At = smem + s * 2 * TILE_BYTES;
    glds_tile(p.A, p.lda, p.R, row0, k0, tid, At);
    glds_tile(p.B, p.ldb, p.Cn, col0, k0, tid, At + TILE_BYTES);
  };

  // per-lane fragment byte offsets inside a tile (row = 16-aligned base + l15)
  // swizzle key of row (w*64 + f*16 + l15) = ((l15>>1) ^ (w*4 + f)) & 7 = s0 ^ f
  const int s0a = ((l15 >> 1) ^ (wn * 4)) & 7;
  const int s0b = ((l15 >> 1) ^ (wm * 4)) & 7;
  const int a_base = (wn * 64 + l15) * 128;  // a-operand = column (B) tile
  const int b_base = (wm * 64 + l15) * 128;  // b-operand = row (A) tile

  auto compute = [&](int s) {
    const char* At = smem + s * 2 * TILE_BYTES;
    const char* Bt = At + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ca = ((g + 4 * kk) ^ s0a) << 4, cb = ((g + 4 * kk) ^ s0b) << 4;
      uint4 af[4], bf[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        af[f] = *reinterpret_cast<const uint4*>(Bt + a_base + f * 16 * 128 + (ca ^ (f << 4)));
        bf[f] = *reinterpret_cast<const uint4*>(At + b_base + f * 16 * 128 + (cb ^ (f << 4)));
      }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = mfma16(af[fn], bf[fm], acc[fm][fn]);
    }
  };

  if (nk > 0) {
    if constexpr (GLDS) {
      stage_glds(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) stage_glds(t + 1, (t + 1) & 1);
        compute(t & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
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

  // ---- epilogue: lane holds C[m][n..n+3] for each (fm, fn) ------------------------------------
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int m = row0 + wm * 64 + fm * 16 + l15;
    if (m >= p.R) continue;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      const int n = col0 + wn * 64 + fn * 16 + g * 4;
      if (n >= p.Cn) continue;
      f32x4_t v = acc[fm][fn];
      if constexpr (F32OUT) {
        float* Cf = reinterpret_cast<float*>(p.C) + (size_t)blockIdx.z * p.R * p.ldc;
        *reinterpret_cast<float4*>(Cf + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        if (p.bias) {
          uint2 bb = *reinterpret_cast<const uint2*>(p.bias + n);
          v[0] += __uint_as_float(bb.x << 16); v[1] += __uint_as_float(bb.x & 0xffff0000u);
          v[2] += __uint_as_float(bb.y << 16); v[3] += __uint_as_float(bb.y & 0xffff0000u);
        }
        if (p.resid) {
          uint2 rr = *reinterpret_cast<const uint2*>(p.resid + (size_t)m * p.ldc + n);
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = o;
      }
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s][i]   (fp32, deterministic split-K finish)
__global__ void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n,
                                     int splits, int accumulate) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float4 s = accumulate ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
  for (int k = 0; k < splits; ++k) {
    float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * n + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(out + i) = s;
}

template <bool TA, bool TB, bool F32OUT, bool GLDS>
int launch(const GemmArgs& a, int splits, hipStream_t st) {
  dim3 grid(a.tiles_r * a.tiles_c, 1, splits);
  gemm_kernel<TA, TB, F32OUT, GLDS><<<grid, 256, 0, st>>>(a);
  return (int)hipGetLastError();
}

}  // namespace

namespace slam {

static int g_gemm_glds = 1;
void gemm_set_glds(int on) { g_gemm_glds = on; }

static int check_dims(int R, int Cn, int Kc, int lda, int ldb, int ldc) {
  if (R <= 0 || Cn <= 0 || Kc <= 0) return -1;
  if ((Cn & 7) || (lda & 7) || (ldb & 7) || (ldc & 3)) return -1;
  return 0;
}

// Y[M,N] = X[M,K] W[N,K]^T (+bias[N]) (+resid[M,N]); bf16 in/out, fp32 accumulate.
int gemm_nt(const bf16_t* X, const bf16_t* W, bf16_t* Y, const bf16_t* bias, const bf16_t* resid, int M,
            int N, int K, hipStream_t st) {
  if (check_dims(M, N, K, K, K, N) || (K & 7)) return -1;
  GemmArgs a{X, W, Y, bias, resid, M, N, K, K, K, N, ((K + BK - 1) / BK) * BK, (M + BM - 1) / BM,
             (N + BN - 1) / BN};
  bool glds = g_gemm_glds && (K % BK == 0) && (N % BN == 0);
  return glds ? launch<false, false, false, true>(a, 1, st) : launch<false, false, false, false>(a, 1, st);
}

// dX[M,K] = dY[M,N] W[N,K] (+resid[M,K]); contraction over N.
int gemm_nn(const bf16_t* dY, const bf16_t* W, bf16_t* dX, const bf16_t* resid, int M, int N, int K,
            hipStream_t st) {
  if (check_dims(M, K, N, N, K, K) || (N & 7)) return -1;
  GemmArgs a{dY, W, dX, nullptr, resid, M, K, N, N, K, K, ((N + BK - 1) / BK) * BK, (M + BM - 1) / BM,
             (K + BN - 1) / BN};
  return launch<false, true, false, false>(a, 1, st);
}

int gemm_tn_splits(int M, int N, int K) {
  int tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN);
  int s = (768 + tiles - 1) / tiles;
  int maxs = (M + 511) / 512;  // at least 8 k-steps per slice
  if (s > maxs) s = maxs;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}
size_t gemm_tn_workspace_bytes(int M, int N, int K) {
  return (size_t)gemm_tn_splits(M, N, K) * N * K * sizeof(float);
}

// dW[N,K] (fp32) (+)= dY[M,N]^T X[M,K]; contraction over M; split-K partials in `ws`.
int gemm_tn(const bf16_t* dY, const bf16_t* X, float* dW, int accumulate, int M, int N, int K, int ldy,
            int ldx, float* ws, hipStream_t st) {
  if (check_dims(N, K, M, ldy, ldx, K) || (N & 7)) return -1;
  int splits = gemm_tn_splits(M, N, K);
  int per = (((M + splits - 1) / splits) + BK - 1) / BK * BK;
  splits = (M + per - 1) / per;
  GemmArgs a{dY, X, ws, nullptr, nullptr, N, K, M, ldy, ldx, K, per, (N + BM - 1) / BM, (K + BN - 1) / BN};
  int e = launch<true, true, true, false>(a, splits, st);
  if (e) return e;
  size_t n = (size_t)N * K;
  reduce_splits_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(ws, dW, n, splits, accumulate);
  return (int)hipGetLastError();
}

}  // namespace slam
