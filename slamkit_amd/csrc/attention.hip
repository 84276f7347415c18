// Causal grouped-query flash attention (head_dim 64 or 128) for gfx950, forward + backward, over a packed
// token axis: row m of qkv[M][(nH+2nKV)*D] attends rows seg_start[m] <= j <= m. Dense [B,T]
// batches are the special case seg_start = (m/T)*T; right padding needs no key mask (a real
// query never sees a later pad key under the causal mask); packed batches pass the segment
// starts derived from position_ids == 0 (flattening collator, hf_dataset.py:61-62).
//
// Replaces Qwen2Attention's softmax_fp32(QKᵀ/8 + mask)·V (site-packages
// transformers/models/qwen2/modeling_qwen2.py:138-172) and its autograd; SURVEY.md §8a T5.
//
// Structure shared by the three MFMA kernels:
//  * scores are produced TRANSPOSED (a-operand = keys, b-operand = queries, or vice versa in dKV)
//    so the contraction index of the following MFMA already sits in the (lane>>4, reg) position
//    of the C fragment and the bf16 P / dS fragment is fed straight back as an MFMA operand;
//  * every 64x64 operand tile arrives by LDS-DMA (global_load_lds_dwordx4, counted vmcnt, ring of
//    stages, ONE barrier per tile, no register staging): a "D image" (16-B chunk swizzle, read
//    with ds_read_b128 when the contraction runs along head_dim) and/or a "T image" (32-B block
//    swizzle, read with ds_read_b64_tr_b16 when the contraction runs along the rows);
//  * per-element masking only on tiles that touch the diagonal, a segment start or the tail;
//  * head_dim D = 64*ND: every operand tile is ND side-by-side 64x64 sub-images (columns 64*dh..),
//    each with the 64-column layouts above, so the D = 128 kernels are the same code with one more
//    loop level (Qwen2.5-1.5B-shaped models).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int IMG = 64 * 128;  // one 64x64 bf16 LDS image
constexpr float NEG_BIG = -1.0e30f;

struct AttnArgs {
  const bf16_t* qkv;   // [M][ldq]
  bf16_t* o;           // [M][nH*D]             (fwd out / bwd in)
  const bf16_t* d_o;   // [M][nH*D]
  bf16_t* dqkv;        // [M][ldq]
  float* lse2;         // [nH][M]  log2-domain logsumexp of scaled scores
  float* dsum;         // [nH][M]  rowsum(dO*O)
  float* dkv_part;     // [2][nH][M][D] fp32 per-q-head dK / dV partials
  const int* seg_start;  // [M]
  const int* seg_end;    // [M]
  const float* rope_cs;  // nullable fp32 [M][D/2]: fold the transpose RoPE rotation into the dq / dk stores
  const float* rope_sn;
  const int* perm;     // nullable: block rank -> q/key tile index, heaviest tiles first (attn_plan_kernel)
  int M, nH, nKV, ldq;
  float scale;         // head_dim^-0.5
};

// DMA one 64x64 tile (rows row0.., clamped to M-1) into an LDS image; 2 x 16 B per thread.
// The per-lane byte offsets are tile-invariant (TileOff, computed once); per tile only the
// wave-uniform base pointer moves. Tiles that cross row M take the clamped slow path.
template <bool TIMG>
struct TileOff {
  uint32_t v[2];
  int ld, tid;
  SLAM_DEVICE void init(int ld_, int tid_) {
    ld = ld_; tid = tid_;
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = off(i, 63);
  }
  SLAM_DEVICE uint32_t off(int i, int maxrow) const {
    int P = i * 256 + tid;
    int row = P >> 3, cs = P & 7;
    int c = TIMG ? ((((cs >> 1) ^ ((row >> 1) & 3)) << 1) | (cs & 1)) : (cs ^ lds_swz_key(row));
    row = row < maxrow ? row : maxrow;
    return (uint32_t)(((size_t)row * ld + c * 8) * sizeof(bf16_t));
  }
};
template <bool TIMG>
SLAM_DEVICE void dma_tile64(const bf16_t* base, const TileOff<TIMG>& to, int row0, int M, int wave, uint32_t img) {
  const bf16_t* tb = base + (size_t)row0 * to.ld;  // wave-uniform
  if (row0 + 64 <= M) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16_sv(tb, to.v[i], __builtin_amdgcn_readfirstlane(img + (uint32_t)(i * 256 + wave * 64) * 16u));
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16_sv(tb, to.off(i, M - 1 - row0), __builtin_amdgcn_readfirstlane(img + (uint32_t)(i * 256 + wave * 64) * 16u));
  }
}
// a-operand fragment from a D image: row = f*16 + l15, head_dim block g + 4*ds
SLAM_DEVICE uint4 frag_direct(const char* img, int f, int l15, int g, int ds) {
  return *reinterpret_cast<const uint4*>(img + lds_tile_off(f * 16 + l15, g + 4 * ds));
}
// a-operand fragment from a T image for contraction step t: column d = fd*16 + l15, rows
// {32t + 4g + r} U {32t + 16 + 4g + r}, r = 0..3 (the order the P / dS b-operand is packed in).
// Lane (l15, g) addresses row 32t + 4g + (l15>>2), columns fd*16 + 4(l15&3)..+3; the 32-B block
// index is XORed with (row>>1)&3 = ((g&1)<<1)|(l15>>3) so a 32-lane group hits 8 distinct windows.
SLAM_DEVICE uint4 frag_tr(const char* img, int fd, int l15, int g, int t) {
  const int k2 = ((g & 1) << 1) | (l15 >> 3);
  const char* p = img + (32 * t + 4 * g + (l15 >> 2)) * 128 + ((fd ^ k2) << 5) + (l15 & 3) * 8;
  uint2 lo = lds_tr_read(p), hi = lds_tr_read(p + 16 * 128);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
SLAM_DEVICE uint4 pack_pair(const f32x4_t& a, const f32x4_t& b) {
  return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                    pack_bf16x2(b[2], b[3]));
}

// ------------------------------------------------------------------------------------------
// Forward. grid (ceil(M/128), nH); wave w owns query rows q0+32w .. +31 (two 16-row fragments).
// Stage = K D-image + V T-image (16 KB per 64 head-dim columns), 3-stage ring at head_dim 64 / 2-stage at 128.
template <int ND>
struct FwdCfg {
  static constexpr int NST = ND == 1 ? 3 : 2;   // ring depth: 48 KB (3 blocks/CU) or 64 KB (2 blocks/CU)
  static constexpr int STAGE = 2 * ND * IMG;
  static constexpr int OCC = ND == 1 ? 3 : 2;
};
template <int ND>
__global__ __launch_bounds__(256, FwdCfg<ND>::OCC) void attn_fwd_kernel(AttnArgs p) {
  constexpr int FWD_NST = FwdCfg<ND>::NST, STG = FwdCfg<ND>::STAGE, D = 64 * ND;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.nH, kvh = h / (p.nH / p.nKV);
  const int slot = blockIdx.x / p.nH;
  const int q0 = (p.perm ? p.perm[slot] : slot) * 128;
  const int qw0 = q0 + wave * 32;
  const int M = p.M, ld = p.ldq;
  const bf16_t* Qb = p.qkv + h * D;
  const bf16_t* Kb = p.qkv + (p.nH + kvh) * D;
  const bf16_t* Vb = p.qkv + (p.nH + p.nKV + kvh) * D;
  const float c2 = p.scale * 1.44269504088896340736f;
  const uint32_t lds0 = lds_addr(smem);

  const int kt_begin = p.seg_start[q0 < M ? q0 : M - 1] / 64;
  const int kt_end = (min(q0 + 127, M - 1)) / 64;
  const int n = kt_end - kt_begin + 1;
  TileOff<false> offD; TileOff<true> offT;
  offD.init(ld, tid); offT.init(ld, tid);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % FWD_NST) * STG);
#pragma unroll
    for (int dh = 0; dh < ND; ++dh) {
      dma_tile64<false>(Kb + dh * 64, offD, (kt_begin + t) * 64, M, wv, st + dh * IMG);
      dma_tile64<true>(Vb + dh * 64, offT, (kt_begin + t) * 64, M, wv, st + (ND + dh) * IMG);
    }
  };
#pragma unroll
  for (int s = 0; s < FWD_NST - 1; ++s)
    if (s < n) issue(s);

  int qrow[2], segs[2];
  uint4 qf[2][2 * ND];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int q = qw0 + j * 16 + l15;
    qrow[j] = q;
    int qc = q < M ? q : M - 1;
    segs[j] = p.seg_start[qc];
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
      qf[j][ds] = *reinterpret_cast<const uint4*>(Qb + (size_t)qc * ld + g * 8 + 32 * ds);
  }
  const int segmax_w = p.seg_start[min(qw0 + 31, M - 1)];  // latest segment start among the wave's rows
  f32x4_t ot[4 * ND][2];
#pragma unroll
  for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
    for (int j = 0; j < 2; ++j) ot[fd][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {NEG_BIG, NEG_BIG}, lsum[2] = {0.f, 0.f};  // running max in raw-score units

  for (int t = 0; t < n; ++t) {
    // 3-deep ring: tile t landed once at most one later tile (4 DMAs per sub-image pair) is in flight
    if (FWD_NST >= 3 && n - 1 - t >= 1) wait_vmcnt<4 * ND>();
    else wait_vmcnt<0>();
    __syncthreads();
    if (t + FWD_NST - 1 < n) issue(t + FWD_NST - 1);
    const char* Ks = smem + (t % FWD_NST) * STG;
    const char* Vs = Ks + ND * IMG;
    const int key0 = (kt_begin + t) * 64;
    if (key0 > qw0 + 31) continue;  // wave-uniform: tile entirely above this wave's diagonal
    f32x4_t st[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 2; ++j) st[f][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        uint4 kf = frag_direct(Ks + (ds >> 1) * IMG, f, l15, g, ds & 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) st[f][j] = mfma16(kf, qf[j][ds], st[f][j]);
      }
    uint4 pb[2][2];
    const bool need_mask = (key0 + 63 > qw0) || (key0 < segmax_w);
    // two separately compiled bodies so the unmasked fast path carries no compare / select at all
    auto softmax_tile = [&](auto mk) {
      constexpr bool MASK = decltype(mk)::value;
      // Deferred running max (threshold 8 in the exp2 domain, P <= 256): the running max of a row is
      // only raised - with the cross-lane reduction, the exp of the correction and the rescale of O -
      // when some element of the tile exceeds it by more than the threshold. The test is lane-local
      // (each lane checks its own 16 scores against the shared max) and wave-uniform via a ballot, so
      // the common tile has no shuffles and no dependent chain through LDS.
      float mloc[2];
      bool grow = false;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (MASK) {
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int key = key0 + f * 16 + g * 4 + r;
              bool ok = (key <= qrow[j]) && (key >= segs[j]);
              st[f][j][r] = ok ? st[f][j][r] : NEG_BIG;
            }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[f][j][r]);
        mloc[j] = mx;
        grow |= (mx - mrun[j]) * c2 > 8.0f;
      }
      if (__any(grow)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float mx = mloc[j];
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float mnew = fmaxf(mrun[j], mx);
          const float alpha = fast_exp2((mrun[j] - mnew) * c2);
          mrun[j] = mnew;
          lsum[j] *= alpha;
#pragma unroll
          for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[fd][j][r] *= alpha;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float mc = mrun[j] * c2;
        float ps = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // masked entries are NEG_BIG: exp2 underflows to 0 unless the whole row is still masked
            // (mrun == NEG_BIG), which the explicit select handles
            float e = fast_exp2(fmaf(st[f][j][r], c2, -mc));
            if constexpr (MASK) e = (st[f][j][r] <= 0.5f * NEG_BIG) ? 0.f : e;
            st[f][j][r] = e;
            ps += e;
          }
        lsum[j] += ps;
        pb[0][j] = pack_pair(st[0][j], st[1][j]);
        pb[1][j] = pack_pair(st[2][j], st[3][j]);
      }
    };
    if (need_mask) softmax_tile(std::true_type{});
    else softmax_tile(std::false_type{});
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) {
        uint4 vf = frag_tr(Vs + (fd >> 2) * IMG, fd & 3, l15, g, t2);
#pragma unroll
        for (int j = 0; j < 2; ++j) ot[fd][j] = mfma16(vf, pb[t2][j], ot[fd][j]);
      }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float l = lsum[j];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int q = qrow[j];
    if (q < M) {
      const float inv = 1.f / l;
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) {
        uint2 o;
        o.x = pack_bf16x2(ot[fd][j][0] * inv, ot[fd][j][1] * inv);
        o.y = pack_bf16x2(ot[fd][j][2] * inv, ot[fd][j][3] * inv);
        *reinterpret_cast<uint2*>(p.o + (size_t)q * (p.nH * D) + h * D + fd * 16 + g * 4) = o;
      }
      if (g == 0 && p.lse2) p.lse2[(size_t)h * M + q] = mrun[j] * c2 + log2f(l);
    }
  }
}

// ------------------------------------------------------------------------------------------
// dQ. grid (ceil(M/64), nH); wave w owns query rows q0+16w .. +15.
// Stage = K D-image + K T-image + V D-image (24 KB per 64 head-dim columns), 6 DMAs per lane per tile and sub-image.
// Ring depth of the two backward kernels: ONE stage. The operand set of a tile is large (dq: 3 images per 64
// head-dim columns, dkv: 4 + the row scalars), so a second stage halves the blocks a CU can hold; the fetch of
// a block is hidden by the MFMA phases of its co-resident blocks instead (head_dim 64: dq 24 KB -> 4 waves/SIMD,
// dkv 33 KB -> 3; head_dim 128: dq 48 KB -> 3 blocks/CU, dkv 65 KB -> 2). Measured against the 2-stage ring:
// backward 167 -> 155 us at head_dim 64 (8 x 1024 tokens, 14 heads), 464 -> 326 us at head_dim 128.
template <int ND> struct BwdCfg { static constexpr int NST = 1; };
template <int ND>
__global__ __launch_bounds__(256, ND == 1 ? 4 : 2) void attn_bwd_dq_kernel(AttnArgs p) {
  constexpr int D = 64 * ND, STG = 3 * ND * IMG, DQ_NST = BwdCfg<ND>::NST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.nH, kvh = h / (p.nH / p.nKV);
  const int slot = blockIdx.x / p.nH;
  const int q0 = (p.perm ? p.perm[slot] : slot) * 64;
  const int qw0 = q0 + wave * 16;
  const int M = p.M, ld = p.ldq;
  const bf16_t* Qb = p.qkv + h * D;
  const bf16_t* Kb = p.qkv + (p.nH + kvh) * D;
  const bf16_t* Vb = p.qkv + (p.nH + p.nKV + kvh) * D;
  const float c2 = p.scale * 1.44269504088896340736f;
  const uint32_t lds0 = lds_addr(smem);

  const int kt_begin = p.seg_start[q0 < M ? q0 : M - 1] / 64;
  const int kt_end = (min(q0 + 63, M - 1)) / 64;
  const int n = kt_end - kt_begin + 1;
  TileOff<false> offD; TileOff<true> offT;
  offD.init(ld, tid); offT.init(ld, tid);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % DQ_NST) * STG);
    const int r0 = (kt_begin + t) * 64;
#pragma unroll
    for (int dh = 0; dh < ND; ++dh) {
      dma_tile64<false>(Kb + dh * 64, offD, r0, M, wv, st + dh * IMG);
      dma_tile64<true>(Kb + dh * 64, offT, r0, M, wv, st + (ND + dh) * IMG);
      dma_tile64<false>(Vb + dh * 64, offD, r0, M, wv, st + (2 * ND + dh) * IMG);
    }
  };
#pragma unroll
  for (int s = 0; s < DQ_NST - 1; ++s)
    if (s < n) issue(s);

  const int q = qw0 + l15;
  const int qc = q < M ? q : M - 1;
  const int seg = p.seg_start[qc];
  const float lse = p.lse2[(size_t)h * M + qc];
  const int segmax_w = p.seg_start[min(qw0 + 15, M - 1)];
  uint4 qf[2 * ND], dof[2 * ND];
  float dsm = 0.f;  // D[q] = sum_d dO[q][d] * O[q][d]: each lane owns a quarter of the d's, 4 lanes per row
#pragma unroll
  for (int ds = 0; ds < 2 * ND; ++ds) {
    qf[ds] = *reinterpret_cast<const uint4*>(Qb + (size_t)qc * ld + g * 8 + 32 * ds);
    dof[ds] = *reinterpret_cast<const uint4*>(p.d_o + (size_t)qc * p.nH * D + h * D + g * 8 + 32 * ds);
    uint4 of = *reinterpret_cast<const uint4*>(p.o + (size_t)qc * p.nH * D + h * D + g * 8 + 32 * ds);
    float x[8], y[8];
    unpack_bf16x8(dof[ds], x);
    unpack_bf16x8(of, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) dsm += x[j] * y[j];
  }
  dsm += __shfl_xor(dsm, 16, 64);
  dsm += __shfl_xor(dsm, 32, 64);
  if (g == 0 && q < M) p.dsum[(size_t)h * M + q] = dsm;  // consumed by the dK/dV kernel (launched after this one)
  f32x4_t dq[4 * ND];
#pragma unroll
  for (int fd = 0; fd < 4 * ND; ++fd) dq[fd] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < n; ++t) {
    if (DQ_NST == 1) {
      __syncthreads();  // every wave is done with the single stage
      issue(t);
    }
    if (DQ_NST >= 3 && n - 1 - t >= 1) wait_vmcnt<6>();
    else wait_vmcnt<0>();
    __syncthreads();
    if (DQ_NST > 1 && t + DQ_NST - 1 < n) issue(t + DQ_NST - 1);
    const char* Ks = smem + (t % DQ_NST) * STG;
    const char* Kt = Ks + ND * IMG;
    const char* Vs = Ks + 2 * ND * IMG;
    const int key0 = (kt_begin + t) * 64;
    if (key0 > qw0 + 15) continue;
    f32x4_t st[4], dp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { st[f] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[f] = st[f]; }
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        st[f] = mfma16(frag_direct(Ks + (ds >> 1) * IMG, f, l15, g, ds & 1), qf[ds], st[f]);
        dp[f] = mfma16(frag_direct(Vs + (ds >> 1) * IMG, f, l15, g, ds & 1), dof[ds], dp[f]);
      }
    const bool need_mask = (key0 + 63 > qw0) || (key0 < segmax_w) || (qw0 + 15 >= M);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pe = fast_exp2(fmaf(st[f][r], c2, -lse));
        if (need_mask) {
          int key = key0 + f * 16 + g * 4 + r;
          bool ok = (key <= q) && (key >= seg) && (q < M);
          pe = ok ? pe : 0.f;
        }
        st[f][r] = pe * (dp[f][r] - dsm);
      }
    uint4 dsb[2] = {pack_pair(st[0], st[1]), pack_pair(st[2], st[3])};
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) dq[fd] = mfma16(frag_tr(Kt + (fd >> 2) * IMG, fd & 3, l15, g, t2), dsb[t2], dq[fd]);
  }
  if (q < M) {
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[fd][r] *= p.scale;
    if (p.rope_cs) {  // transpose rotation: d(pre-RoPE q); fragments fd and fd + 2ND hold d and d + D/2
#pragma unroll
      for (int fd = 0; fd < 2 * ND; ++fd) {
        const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cs + (size_t)q * (D / 2) + fd * 16 + g * 4);
        const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sn + (size_t)q * (D / 2) + fd * 16 + g * 4);
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = dq[fd][r], b = dq[fd + 2 * ND][r];
          dq[fd][r] = a * cc[r] + b * ss[r];
          dq[fd + 2 * ND][r] = b * cc[r] - a * ss[r];
        }
      }
    }
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) {
      uint2 o;
      o.x = pack_bf16x2(dq[fd][0], dq[fd][1]);
      o.y = pack_bf16x2(dq[fd][2], dq[fd][3]);
      *reinterpret_cast<uint2*>(p.dqkv + (size_t)q * ld + h * D + fd * 16 + g * 4) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// dK / dV per query head. grid (ceil(M/64), nH); wave w owns keys k0+16w .. +15; fp32 partials
// dkv_part[0|1][h][m][64] are summed over the heads of a KV group by attn_dkv_reduce_kernel.
// Stage = Q D/T images + dO D/T images (32 KB) + lse2 / dsum / seg_start of the 64 query rows
// (3 x 256 B, by 4-byte LDS-DMA), 2-stage ring, 9 DMAs per lane per tile.
template <int ND>
__global__ __launch_bounds__(256, ND == 1 ? 3 : 2) void attn_bwd_dkv_kernel(AttnArgs p) {
  constexpr int D = 64 * ND, DKV_STAGE = 4 * ND * IMG + 1024, DKV_NST = BwdCfg<ND>::NST;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.nH, kvh = h / (p.nH / p.nKV);
  const int slot = blockIdx.x / p.nH;
  const int k0 = (p.perm ? p.perm[slot] : slot) * 64;
  const int M = p.M, ld = p.ldq, ldo = p.nH * D;
  const bf16_t* Qb = p.qkv + h * D;
  const bf16_t* Kb = p.qkv + (p.nH + kvh) * D;
  const bf16_t* Vb = p.qkv + (p.nH + p.nKV + kvh) * D;
  const bf16_t* dOb = p.d_o + h * D;
  const float c2 = p.scale * 1.44269504088896340736f;
  const uint32_t lds0 = lds_addr(smem);

  const int qt_begin = k0 / 64;
  const int qt_end = (p.seg_end[min(k0 + 63, M - 1)] - 1) / 64;
  const int n = qt_end - qt_begin + 1;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  TileOff<false> qD, oD; TileOff<true> qT, oT;
  qD.init(ld, tid); qT.init(ld, tid); oD.init(ldo, tid); oT.init(ldo, tid);
  auto issue = [&](int t) {
    const uint32_t st = lds0 + (uint32_t)((t % DKV_NST) * DKV_STAGE);
    const int r0 = (qt_begin + t) * 64;
#pragma unroll
    for (int dh = 0; dh < ND; ++dh) {
      dma_tile64<false>(Qb + dh * 64, qD, r0, M, wv, st + dh * IMG);
      dma_tile64<true>(Qb + dh * 64, qT, r0, M, wv, st + (ND + dh) * IMG);
      dma_tile64<false>(dOb + dh * 64, oD, r0, M, wv, st + (2 * ND + dh) * IMG);
      dma_tile64<true>(dOb + dh * 64, oT, r0, M, wv, st + (3 * ND + dh) * IMG);
    }
    // per-row scalars: wave 0 -> lse2, 1 -> dsum, 2 -> seg_start, 3 -> spare slot (keeps the DMA count uniform)
    const int row = min(r0 + lane, M - 1);
    const void* src = wv == 0 ? (const void*)(p.lse2 + (size_t)h * M + row)
                    : wv == 1 ? (const void*)(p.dsum + (size_t)h * M + row)
                              : (const void*)(p.seg_start + row);
    glds4(src, __builtin_amdgcn_readfirstlane(st + 4 * ND * IMG + (uint32_t)wv * 256u));
  };
  if (DKV_NST > 1 && n > 0) issue(0);

  const int key = k0 + wave * 16 + l15;
  const int kc = key < M ? key : M - 1;
  uint4 kf[2 * ND], vf[2 * ND];
#pragma unroll
  for (int ds = 0; ds < 2 * ND; ++ds) {
    kf[ds] = *reinterpret_cast<const uint4*>(Kb + (size_t)kc * ld + g * 8 + 32 * ds);
    vf[ds] = *reinterpret_cast<const uint4*>(Vb + (size_t)kc * ld + g * 8 + 32 * ds);
  }
  f32x4_t dk[4 * ND], dv[4 * ND];
#pragma unroll
  for (int fd = 0; fd < 4 * ND; ++fd) { dk[fd] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[fd] = dk[fd]; }

  for (int t = 0; t < n; ++t) {
    if (DKV_NST == 1) {
      __syncthreads();
      issue(t);
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (DKV_NST > 1 && t + 1 < n) issue(t + 1);
    const char* Qs = smem + (t % DKV_NST) * DKV_STAGE;
    const char* Qt = Qs + ND * IMG;
    const char* dOs = Qs + 2 * ND * IMG;
    const char* dOt = Qs + 3 * ND * IMG;
    const float* lse_s = reinterpret_cast<const float*>(Qs + 4 * ND * IMG);
    const float* dsm_s = lse_s + 64;
    const int* seg_s = reinterpret_cast<const int*>(lse_s + 128);
    const int qbase = (qt_begin + t) * 64;
    const int kw0 = k0 + wave * 16;
    if (qbase + 63 < kw0) continue;  // no query of the tile can see this wave's keys
    f32x4_t s[4], dp[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) { s[jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[jq] = s[jq]; }
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        s[jq] = mfma16(frag_direct(Qs + (ds >> 1) * IMG, jq, l15, g, ds & 1), kf[ds], s[jq]);
        dp[jq] = mfma16(frag_direct(dOs + (ds >> 1) * IMG, jq, l15, g, ds & 1), vf[ds], dp[jq]);
      }
    // lane holds (q = qbase + jq*16 + 4g + r, key); mask only on diagonal / segment-boundary / tail tiles
    const bool need_mask = (kw0 + 15 > qbase) || (kw0 < seg_s[63]) || (qbase + 63 >= M);
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const float4 l4 = *reinterpret_cast<const float4*>(lse_s + jq * 16 + g * 4);
      const float4 d4 = *reinterpret_cast<const float4*>(dsm_s + jq * 16 + g * 4);
      const int4 s4 = *reinterpret_cast<const int4*>(seg_s + jq * 16 + g * 4);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
      const int sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pe = fast_exp2(fmaf(s[jq][r], c2, -lv[r]));
        if (need_mask) {
          int qq = qbase + jq * 16 + g * 4 + r;
          bool ok = (key <= qq) && (key >= sv[r]) && (qq < M);
          pe = ok ? pe : 0.f;
        }
        s[jq][r] = pe;
        dp[jq][r] = pe * (dp[jq][r] - dvv[r]);
      }
    }
    uint4 pb[2] = {pack_pair(s[0], s[1]), pack_pair(s[2], s[3])};
    uint4 dsb[2] = {pack_pair(dp[0], dp[1]), pack_pair(dp[2], dp[3])};
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) {
        dv[fd] = mfma16(frag_tr(dOt + (fd >> 2) * IMG, fd & 3, l15, g, t2), pb[t2], dv[fd]);
        dk[fd] = mfma16(frag_tr(Qt + (fd >> 2) * IMG, fd & 3, l15, g, t2), dsb[t2], dk[fd]);
      }
  }
  if (key < M) {
    float* dkp = p.dkv_part + ((size_t)h * M + key) * D;
    float* dvp = p.dkv_part + ((size_t)(p.nH + h) * M + key) * D;
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) {
      *reinterpret_cast<float4*>(dkp + fd * 16 + g * 4) =
          make_float4(dk[fd][0] * p.scale, dk[fd][1] * p.scale, dk[fd][2] * p.scale, dk[fd][3] * p.scale);
      *reinterpret_cast<float4*>(dvp + fd * 16 + g * 4) = make_float4(dv[fd][0], dv[fd][1], dv[fd][2], dv[fd][3]);
    }
  }
}

// dqkv[m][K head kvh / V head kvh] = bf16( sum over the group's query heads of the partials ); a thread
// owns d = 4c..4c+3 and its rotate-half partner d + D/2, so dK can be rotated back in the same pass.
template <int ND>
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnArgs p) {
  constexpr int D = 64 * ND, HALF = D / 2, CPR = D / 8;  // CPR threads per (m, kv head)
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // (which, m, kvh, c < CPR)
  const int grp = p.nH / p.nKV;
  size_t total = (size_t)2 * p.M * p.nKV * CPR;
  if (idx >= total) return;
  int c = idx % CPR;
  size_t r = idx / CPR;
  int kvh = r % p.nKV; r /= p.nKV;
  int m = r % p.M;
  int which = (int)(r / p.M);
  float4 a = make_float4(0, 0, 0, 0), b = a;
  for (int i = 0; i < grp; ++i) {
    int h = kvh * grp + i;
    const float* src = p.dkv_part + (((size_t)which * p.nH + h) * p.M + m) * D + c * 4;
    float4 v = *reinterpret_cast<const float4*>(src), w = *reinterpret_cast<const float4*>(src + HALF);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
  }
  if (which == 0 && p.rope_cs) {
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cs + (size_t)m * HALF + c * 4);
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sn + (size_t)m * HALF + c * 4);
    float4 x = a, y = b;
    a = make_float4(x.x * c4.x + y.x * s4.x, x.y * c4.y + y.y * s4.y, x.z * c4.z + y.z * s4.z, x.w * c4.w + y.w * s4.w);
    b = make_float4(y.x * c4.x - x.x * s4.x, y.y * c4.y - x.y * s4.y, y.z * c4.z - x.z * s4.z, y.w * c4.w - x.w * s4.w);
  }
  uint2 o1, o2;
  o1.x = pack_bf16x2(a.x, a.y); o1.y = pack_bf16x2(a.z, a.w);
  o2.x = pack_bf16x2(b.x, b.y); o2.y = pack_bf16x2(b.z, b.w);
  int col = (p.nH + (which ? p.nKV : 0) + kvh) * D + c * 4;
  *reinterpret_cast<uint2*>(p.dqkv + (size_t)m * p.ldq + col) = o1;
  *reinterpret_cast<uint2*>(p.dqkv + (size_t)m * p.ldq + col + HALF) = o2;
}

// Longest-processing-time-first block order. Causal tiles differ 1:16 in work; in launch order the
// heavy tiles of the last sequences start last and the chip drains half empty (60 % schedule
// efficiency at 512 block slots in simulation, 97 % with heaviest-first across all heads).
// plan = [ fwd perm (128-row q tiles) | dq perm (64-row q tiles) | dkv perm (64-row key tiles) ].
__global__ void attn_plan_kernel(const int* __restrict__ seg_s, const int* __restrict__ seg_e, int M,
                                 int* __restrict__ plan) {
  const int nf = (M + 127) / 128, nq = (M + 63) / 64;
  const int total = nf + 2 * nq;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int type, i, n, base;
  if (t < nf) { type = 0; i = t; n = nf; base = 0; }
  else if (t < nf + nq) { type = 1; i = t - nf; n = nq; base = nf; }
  else { type = 2; i = t - nf - nq; n = nq; base = nf + nq; }
  auto work = [&](int j) -> int {
    if (type == 0) { int q0 = j * 128; return min(q0 + 127, M - 1) / 64 - seg_s[q0] / 64 + 1; }
    if (type == 1) { int q0 = j * 64; return min(q0 + 63, M - 1) / 64 - seg_s[q0] / 64 + 1; }
    int k0 = j * 64;
    return (seg_e[min(k0 + 63, M - 1)] - 1) / 64 - k0 / 64 + 1;
  };
  const int w = work(i);
  int rank = 0;
  for (int j = 0; j < n; ++j) {
    int wj = work(j);
    rank += (wj > w) || (wj == w && j < i);
  }
  plan[base + rank] = i;
}

}  // namespace

namespace slam {

size_t attn_plan_ints(int M) { return (size_t)((M + 127) / 128 + 2 * ((M + 63) / 64)); }
int attn_plan(const int* seg_start, const int* seg_end, int M, int* plan, hipStream_t st) {
  int total = (int)attn_plan_ints(M);
  attn_plan_kernel<<<(total + 255) / 256, 256, 0, st>>>(seg_start, seg_end, M, plan);
  return (int)hipGetLastError();
}

template <int ND>
static int set_lds_attrs() {
  static bool done = false;
  if (done) return 0;
  hipError_t e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<ND>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          FwdCfg<ND>::NST * FwdCfg<ND>::STAGE);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<ND>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          BwdCfg<ND>::NST * 3 * ND * IMG);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<ND>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          BwdCfg<ND>::NST * (4 * ND * IMG + 1024));
  if (e != hipSuccess) return (int)e;
  done = true;
  return 0;
}

template <int ND>
static int attn_fwd_nd(AttnArgs a, hipStream_t st) {
  if (int e = set_lds_attrs<ND>()) return e;
  attn_fwd_kernel<ND><<<((a.M + 127) / 128) * a.nH, 256, FwdCfg<ND>::NST * FwdCfg<ND>::STAGE, st>>>(a);
  return (int)hipGetLastError();
}

template <int ND>
static int attn_bwd_nd(AttnArgs a, const int* plan, hipStream_t st) {
  if (int e = set_lds_attrs<ND>()) return e;
  const int M = a.M, nH = a.nH;
  const int nf = (M + 127) / 128, nq = (M + 63) / 64;
  a.perm = plan ? plan + nf : nullptr;
  attn_bwd_dq_kernel<ND><<<nq * nH, 256, BwdCfg<ND>::NST * 3 * ND * IMG, st>>>(a);
  a.perm = plan ? plan + nf + nq : nullptr;
  attn_bwd_dkv_kernel<ND><<<nq * nH, 256, BwdCfg<ND>::NST * (4 * ND * IMG + 1024), st>>>(a);
  size_t total = (size_t)2 * M * a.nKV * (8 * ND);
  attn_dkv_reduce_kernel<ND><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a);
  return (int)hipGetLastError();
}

int attn_fwd(const bf16_t* qkv, bf16_t* o, float* lse2, const int* seg_start, const int* plan, int M, int nH,
             int nKV, int head_dim, hipStream_t st) {
  if ((head_dim != 64 && head_dim != 128) || nH % nKV) return -1;
  AttnArgs a{};
  a.qkv = qkv; a.o = o; a.lse2 = lse2; a.seg_start = seg_start; a.perm = plan;
  a.M = M; a.nH = nH; a.nKV = nKV; a.ldq = (nH + 2 * nKV) * head_dim; a.scale = 1.0f / sqrtf((float)head_dim);
  return head_dim == 64 ? attn_fwd_nd<1>(a, st) : attn_fwd_nd<2>(a, st);
}

size_t attn_bwd_workspace_bytes(int M, int nH, int head_dim) { return (size_t)2 * nH * M * head_dim * sizeof(float); }

int attn_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse2, float* dsum,
             bf16_t* dqkv, float* dkv_part, const int* seg_start, const int* seg_end, const int* plan,
             const float* rope_cs, const float* rope_sn, int M, int nH, int nKV, int head_dim, hipStream_t st) {
  if ((head_dim != 64 && head_dim != 128) || nH % nKV) return -1;
  AttnArgs a{};
  a.qkv = qkv; a.o = const_cast<bf16_t*>(o); a.d_o = d_o; a.dqkv = dqkv;
  a.lse2 = const_cast<float*>(lse2); a.dsum = dsum; a.dkv_part = dkv_part;
  a.seg_start = seg_start; a.seg_end = seg_end;
  a.rope_cs = rope_cs; a.rope_sn = rope_sn;
  a.M = M; a.nH = nH; a.nKV = nKV; a.ldq = (nH + 2 * nKV) * head_dim; a.scale = 1.0f / sqrtf((float)head_dim);
  return head_dim == 64 ? attn_bwd_nd<1>(a, plan, st) : attn_bwd_nd<2>(a, plan, st);
}

}  // namespace slam
