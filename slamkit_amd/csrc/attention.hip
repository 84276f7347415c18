// Causal grouped-query flash attention (head_dim 64 or 128) for gfx950, forward + backward, over a packed
// token axis: row m of qkv[M][(nH+2nKV)*D] attends rows seg_start[m] <= j <= m. The QUERY columns of qkv arrive
// PRE-SCALED by head_dim^-0.5 * log2(e) (folded into their RoPE tables by the QKV projection's epilogue: one
// rounding), so q.k is the score in the exp2 domain. Dense [B,T]
// batches are the special case seg_start = (m/T)*T; right padding needs no key mask (a real
// query never sees a later pad key under the causal mask); packed batches pass the segment
// starts derived from position_ids == 0 (flattening collator, hf_dataset.py:61-62).
//
// Replaces Qwen2Attention's softmax_fp32(QKᵀ/8 + mask)·V with repeat_kv (site-packages
// transformers/models/qwen2/modeling_qwen2.py:138-172) and its autograd; SURVEY.md §8a T5.
//
// Structure shared by the three MFMA kernels (round 3 rewrite):
//  * scores are produced TRANSPOSED (a-operand = keys, b-operand = queries, or vice versa in dKV)
//    so the contraction index of the following MFMA already sits in the (lane>>4, reg) position
//    of the C fragment and the bf16 P / dS fragment is fed straight back as an MFMA operand;
//  * every 64x64 operand tile arrives by LDS-DMA (global_load_lds_dwordx4, counted vmcnt, ring of
//    stages, ONE barrier per tile, no register staging) as ONE "unified image": rows of 128 B whose
//    16-byte chunks are XOR-swizzled with f((row>>1)&7), f(k) = 2(k&3) ^ 5(k>>2). The same image is
//    read conflict-free BOTH ways: with ds_read_b128 when the contraction runs along head_dim
//    ("D fragments": 16 rows x one chunk per 16-lane group) and with ds_read_b64_tr_b16 when it
//    runs along the rows ("T fragments": 8 rows x 32 B per 32-lane group). Round 2 kept two
//    differently swizzled copies (K twice in dQ, Q and dO twice in dK/dV): half the LDS and half
//    the DMA bytes per tile now, which is what pays for the 3-stage rings of the backward kernels;
//  * all LDS addressing inside a tile is one base VGPR per fragment kind (2 D + 4 T, re-based once
//    per tile) plus instruction immediates; masking is branch-free and only compiled into the
//    tile bodies that touch the diagonal, a segment start or the tail (wave-uniform choice);
//  * head_dim D = 64*ND: every operand tile is ND side-by-side 64-column images.
//  * dK/dV: one block owns a 64-key tile of a KV head and walks ALL query heads of its GQA group
//    (the K/V fragments stay in registers, dK/dV accumulate across the group in registers): no
//    per-query-head fp32 slabs. Long key tiles are cut into at most ATTN_NCH_MAX query-range chunks
//    (load balance under the causal mask); the chunks' fp32 partials are summed in chunk order by
//    attn_dkv_reduce_kernel, which also rotates dK back (transpose RoPE) - deterministic, no atomics.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int IMG = 64 * 128;  // one 64x64 bf16 LDS image
constexpr float NEG_BIG = -1.0e30f;  // masked score
constexpr float M_INIT = -1.0e29f;   // initial running max: > NEG_BIG, so exp2((NEG_BIG - M_INIT) * c) == 0 and a
                                     // row that is fully masked in its first tiles needs no select
constexpr int NCH_MAX = 4;
constexpr float LN2 = 0.69314718055994530942f;

struct AttnArgs {
  const bf16_t* qkv;   // [M][ldq]
  bf16_t* o;           // [M][nH*D]             (fwd out / bwd in)
  const bf16_t* d_o;   // [M][nH*D]
  bf16_t* dqkv;        // [M][ldq]
  float* lse2;         // [nH][M]  log2-domain logsumexp of scaled scores
  float* ndsum;        // [nH][M]  -rowsum(dO*O)
  float* nlse;         // [nH][M]  -lse2 (written by the dQ kernel: the dK/dV kernel starts its score accumulators there)
  float* dkv_part;     // [NCH_MAX][2][nKV][M][D] fp32 dK / dV partials per query-range chunk
  const int* seg_start;  // [M]
  const int* seg_end;    // [M]
  const float* rope_cs;  // nullable fp32 [M][D/2]: fold the transpose RoPE rotation into the dq / dk stores
  const float* rope_sn;
  const int* perm;     // nullable: block rank -> q tile index, heaviest tiles first; dK/dV: item list
  int M, nH, nKV, ldq, nch, prio;
  float scale;         // head_dim^-0.5
};

typedef __attribute__((address_space(3))) const char* lds_cptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) const u32x4_t* lds_u4ptr_t;
SLAM_DEVICE lds_cptr_t lds_p(uint32_t a) { return (lds_cptr_t)(uintptr_t)a; }

SLAM_DEVICE int uni_f(int k) { return ((k & 3) << 1) ^ ((k >> 2) * 5); }

// DMA one 64x64 tile (rows row0.., clamped to M-1) into a unified LDS image; 2 x 16 B per thread.
// The per-lane byte offsets are tile-invariant (TileOff, computed once); per tile only the
// wave-uniform base pointer moves. Tiles that cross row M take the clamped slow path.
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
    int row = P >> 3, slot = P & 7;
    int c = slot ^ uni_f((row >> 1) & 7);  // the swizzle sits on the SOURCE address: the LDS side of a DMA is lane-linear
    row = row < maxrow ? row : maxrow;
    return (uint32_t)(((size_t)row * ld + c * 8) * sizeof(bf16_t));
  }
};
// one 64x64 tile whose first row is at `tb` (wave-uniform) -> image at LDS address `dst` (this wave's 1 KB slice of each
// 4 KB half); maxrow >= 63: every row exists (tile-invariant offsets), else rows are clamped to maxrow
template <bool FULL>
SLAM_DEVICE void dma_tile64(const bf16_t* tb, const TileOff& to, int maxrow, uint32_t dst) {
#pragma unroll
  for (int i = 0; i < 2; ++i) glds16_m0(tb, FULL ? to.v[i] : to.off(i, maxrow), dst + (uint32_t)(i * 4096));
}

// Per-lane fragment offsets inside a unified image (tile-invariant); TileAddr = the same re-based on a stage, once per
// tile, kept opaque so that every fragment read is ONE ds_read with an immediate (image, 16-row group, contraction step).
struct FragOff {
  uint32_t d[2], t[4];
  SLAM_DEVICE void init(int l15, int g) {
    // D fragment (ds_read_b128): row 16f + l15, head_dim chunk g + 4ds; key(row) = f(l15>>1) for every f
    const int fk = uni_f(l15 >> 1);
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) d[ds] = (uint32_t)(l15 * 128 + (((g + 4 * ds) ^ fk) << 4));
    // T fragment (2 x ds_read_b64_tr_b16): lane (l15, g) addresses row 32t + 4g + (l15>>2) [+16], columns
    // fd*16 + 4(l15&3)..+3 = chunk 2fd + ((l15&3)>>1), byte (l15&1)*8; key(row) = f(2g + (l15>>3)) for every t, both halves
    const int x = ((l15 & 3) >> 1) ^ uni_f(2 * g + (l15 >> 3));
#pragma unroll
    for (int fd = 0; fd < 4; ++fd)
      t[fd] = (uint32_t)((4 * g + (l15 >> 2)) * 128 + (((2 * fd) ^ x) << 4) + (l15 & 1) * 8);
  }
};
struct TileAddr {
  uint32_t d[2], t[4];
  SLAM_DEVICE void set(const FragOff& fo, uint32_t stage_base) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { d[i] = stage_base + fo.d[i]; asm volatile("" : "+v"(d[i])); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { t[i] = stage_base + fo.t[i]; asm volatile("" : "+v"(t[i])); }
  }
  // a-operand from image `img` (byte offset inside the stage): rows 16f.., head_dim block g + 4ds (contraction along head_dim)
  SLAM_DEVICE uint4 D(int img, int f, int ds) const {
    const u32x4_t v = *(lds_u4ptr_t)(lds_p(d[ds]) + (img + f * 2048));
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
  // a-operand for contraction step t2 (rows 32 t2 .. +31 of the tile), output column block fd (contraction along the rows):
  // rows {32t + 4g + r} U {32t + 16 + 4g + r}, r = 0..3 - the order the P / dS b-operand is packed in
  SLAM_DEVICE uint4 T(int img, int fd, int t2) const {
    lds_cptr_t q = lds_p(t[fd]) + (img + t2 * 4096);
    typedef __attribute__((address_space(3))) s16x4_t* trp_t;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)q);
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(q + 2048));
    uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
  }
};

SLAM_DEVICE uint4 pack_pair(const f32x4_t& a, const f32x4_t& b) {
  return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                    pack_bf16x2(b[2], b[3]));
}
// s_waitcnt vmcnt(0) the compiler can see: after it the waitcnt pass knows that every load it issued itself (the
// register-resident Q / dO / K / V rows of a block) has landed and inserts no vmcnt wait of its own inside the tile loop -
// one there would also wait for the hand-counted LDS-DMA of the NEXT tiles (round 2's forward had exactly that: a
// compiler vmcnt(0) at the first use of the Q fragments in every iteration, i.e. no prefetch at all).
SLAM_DEVICE void wait_all_loads_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// XCD-grouped block -> (tile rank, head): the G query heads of a (tile, KV head) pair run on ONE XCD (block b runs on
// XCD b % 8), so their K/V tiles are fetched into one L2 instead of up to eight.
struct BlockItem { int slot, h, kvh; bool valid; };
SLAM_DEVICE BlockItem block_item(int b, int ntile, int nH, int nKV) {
  const int G = nH / nKV;
  const int seq = b >> 3;
  const int cid = (seq / G) * 8 + (b & 7);
  BlockItem it;
  it.valid = cid < ntile * nKV;
  it.kvh = cid % nKV;
  it.slot = cid / nKV;
  it.h = it.kvh * G + seq % G;
  return it;
}
// Wave priority by LPT rank ("attn_prio"): the blocks of a launch differ 1:16 in length and the longest one IS the
// critical path when three blocks share a CU - the heaviest quarter of the items runs at s_setprio 3, the lightest at 0,
// so a long block is slowed less by its co-resident short ones.
SLAM_DEVICE void set_rank_prio(int rank, int total) {
  const int q = (4 * rank) / (total > 0 ? total : 1);
  if (q <= 0) __builtin_amdgcn_s_setprio(3);
  else if (q == 1) __builtin_amdgcn_s_setprio(2);
  else if (q == 2) __builtin_amdgcn_s_setprio(1);
}
inline int item_grid(int ntile, int nH, int nKV) { return ((ntile * nKV + 7) / 8) * 8 * (nH / nKV); }

// ------------------------------------------------------------------------------------------
// Forward. wave w owns query rows q0+32w .. +31 (two 16-row fragments) of a 128-row tile.
// Stage = K image + V image (16 KB per 64 head-dim columns), 3-stage ring at head_dim 64 / 2-stage at 128.
template <int ND>
struct FwdCfg {
  static constexpr int NST = ND == 1 ? 3 : 2;   // ring depth: 48 KB (3 blocks/CU) or 64 KB (2 blocks/CU)
  static constexpr int STAGE = 2 * ND * IMG;
  static constexpr int OCC = ND == 1 ? 3 : 2;
};
template <int ND>
__global__ __launch_bounds__(256, FwdCfg<ND>::OCC) void attn_fwd_kernel(AttnArgs p) {
  constexpr int NST = FwdCfg<ND>::NST, STG = FwdCfg<ND>::STAGE, D = 64 * ND;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.M, ld = p.ldq;
  const BlockItem bi = block_item(blockIdx.x, (M + 127) / 128, p.nH, p.nKV);
  if (!bi.valid) return;
  const int h = bi.h, kvh = bi.kvh;
  const int q0 = (p.perm ? p.perm[bi.slot] : bi.slot) * 128;
  if (p.prio && p.perm) set_rank_prio(bi.slot, (M + 127) / 128);
  // the wave index as an SGPR: everything derived from it (the wave's first query row, "this tile is above my diagonal",
  // "this tile needs masks") is then a scalar compare + s_cbranch instead of v_cmp + exec-mask save / restore around the body
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int qw0 = q0 + wv * 32;
  const bf16_t* Qb = p.qkv + h * D;
  const bf16_t* Kb = p.qkv + (p.nH + kvh) * D;
  const bf16_t* Vb = p.qkv + (p.nH + p.nKV + kvh) * D;
  const float c2 = p.scale * 1.44269504088896340736f;
  const uint32_t lds0 = lds_addr(smem);

  const int kt_begin = p.seg_start[q0 < M ? q0 : M - 1] / 64;
  const int kt_end = (min(q0 + 127, M - 1)) / 64;
  const int n = kt_end - kt_begin + 1;
  TileOff off;
  off.init(ld, tid);
  FragOff fo;
  fo.init(l15, g);
  // running tile pointers (wave-uniform): one 64-bit add per operand per tile
  const size_t tstep = (size_t)64 * ld;
  const bf16_t* kp = Kb + (size_t)kt_begin * tstep;
  const bf16_t* vp = Vb + (size_t)kt_begin * tstep;
  int irow = kt_begin * 64;
  const uint32_t wdst = lds0 + (uint32_t)wv * 1024u;
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const uint32_t st = wdst + (uint32_t)(stage * STG);
    if (irow + 64 <= M) {  // ONE wave-uniform branch per tile
#pragma unroll
      for (int dh = 0; dh < ND; ++dh) {
        dma_tile64<true>(kp + dh * 64, off, 63, st + dh * IMG);
        dma_tile64<true>(vp + dh * 64, off, 63, st + (ND + dh) * IMG);
      }
    } else {
      const int mr = M - 1 - irow;
#pragma unroll
      for (int dh = 0; dh < ND; ++dh) {
        dma_tile64<false>(kp + dh * 64, off, mr, st + dh * IMG);
        dma_tile64<false>(vp + dh * 64, off, mr, st + (ND + dh) * IMG);
      }
    }
    kp += tstep; vp += tstep; irow += 64;
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
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
  // Instruction diet of the tile loop (a SIMD issues about one instruction per 5-6 cycles in this mix, so at long
  // sequences the loop is bound by its instruction COUNT - measured, tools/probes/ubench.hip): the queries arrive
  // pre-scaled by scale*log2(e) and the score accumulators START at -m (the running max, as a persistent register quad per
  // row block), so a score leaves the matrix pipe as (s - m) in the exp2 domain and the probability is ONE v_exp_f32 - no
  // fma, no per-tile multiply of the max. (Row sums stay fp32 adds of the unrounded probabilities: summing them on the
  // matrix pipe as well - a ones fragment times P^T - saves 20 instructions per tile but sums the bf16-ROUNDED values, which
  // moved lse2 by 2e-4 relative and the 200-step loss curve by 0.3 % at single steps; measured and taken back out.)
  f32x4_t ot[4 * ND][2], negm[2];
  float lsum[2] = {0.f, 0.f};
  // the running max moves when the tile's row max exceeds thr: "any visible key" until the row has seen its first one
  // (m becomes the true row max then), m + 8 afterwards - ONE compare per row block and tile (round 5: a flag, two compares, selects)
  float thr[2] = {0.5f * NEG_BIG, 0.5f * NEG_BIG};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    negm[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};  // m = 0 until the row has seen its first visible key ("started")
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) ot[fd][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  wait_all_loads_visible();

  int stage = 0, istage = (NST - 1) % NST;
  for (int t = 0; t < n; ++t) {
    // 3-deep ring: tile t landed once at most one later tile (4 DMAs per 64 head-dim columns) is in flight
    if (NST >= 3 && t + 1 < n) wait_vmcnt<4 * ND>();
    else wait_vmcnt<0>();
    __syncthreads();
    if (t + NST - 1 < n) issue(istage);
    istage = istage + 1 == NST ? 0 : istage + 1;
    const uint32_t sb = lds0 + (uint32_t)(stage * STG);
    stage = stage + 1 == NST ? 0 : stage + 1;
    const int key0 = (kt_begin + t) * 64;
    if (key0 > qw0 + 31) continue;  // wave-uniform: tile entirely above this wave's diagonal
    TileAddr ta;
    ta.set(fo, sb);
    f32x4_t st[4][2];
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        uint4 kf = ta.D((ds >> 1) * IMG, f, ds & 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) st[f][j] = mfma16(kf, qf[j][ds], ds == 0 ? negm[j] : st[f][j]);
      }
    if ((key0 + 63 > qw0) || (key0 < segmax_w)) {  // wave-uniform: diagonal or segment-boundary tile
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int hi = qrow[j] - key0 - 4 * g, lo = segs[j] - key0 - 4 * g;  // key f*16 + 4g + r visible iff lo <= 16f + r <= hi
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (f * 16 + r <= hi) & (f * 16 + r >= lo);
            st[f][j][r] = ok ? st[f][j][r] : NEG_BIG;
          }
      }
    }
    // Deferred running max: the scores are already relative to m; m moves - with the cross-lane reduction, the exp of the
    // correction, the rescale of O and l and the shift of this tile's scores - only when a score exceeds it by more than
    // 8 (P <= 256), or on the first tile in which the row sees a key at all (then m becomes the true row max: a row
    // whose scores all sit far below zero must not underflow against m = 0). The test is lane-local and wave-uniform
    // via a ballot, so the common tile has no shuffles and no dependent chain.
    float mloc[2];
    bool grow = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float mx = NEG_BIG;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[f][j][r]);
      mloc[j] = mx;
      grow |= mx > thr[j];
    }
    if (__any(grow)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float mx = mloc[j];
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx > thr[j]) {  // this row's max moves (the same decision in the four lanes of a row): m += mx
          const float shift = -mx;                                     // m_old - m_new
          const float alpha = thr[j] > 0.f ? fast_exp2(shift) : 1.f;  // (nothing accumulated yet before the first key: exp2(-mx) may be inf)
          thr[j] = 8.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) negm[j][r] += shift;
          lsum[j] *= alpha;
#pragma unroll
          for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[fd][j][r] *= alpha;
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[f][j][r] += shift;
        }
      }
    }
    // the V^T fragments of the first contraction half are requested BEFORE the exp / pack arithmetic and land under it
    // (left to itself the compiler sinks every fragment read to just before its MFMA pair: eight exposed LDS latencies)
    uint4 vf0[4 * ND];
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) vf0[fd] = ta.T((ND + (fd >> 2)) * IMG, fd & 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    uint4 pb[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float ps = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp2(st[f][j][r]);  // masked entries (NEG_BIG) underflow to exactly 0
          st[f][j][r] = e;
          ps += e;
        }
      lsum[j] += ps;
      pb[0][j] = pack_pair(st[0][j], st[1][j]);
      pb[1][j] = pack_pair(st[2][j], st[3][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    uint4 vf1[4 * ND];  // second half: requested before the first half's MFMAs
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) vf1[fd] = ta.T((ND + (fd >> 2)) * IMG, fd & 3, 1);
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
      for (int j = 0; j < 2; ++j) ot[fd][j] = mfma16(vf0[fd], pb[0][j], ot[fd][j]);
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
      for (int j = 0; j < 2; ++j) ot[fd][j] = mfma16(vf1[fd], pb[1][j], ot[fd][j]);
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
      if (g == 0 && p.lse2) p.lse2[(size_t)h * M + q] = log2f(l) - negm[j][0];
    }
  }
}

// ------------------------------------------------------------------------------------------
// dQ. wave w owns query rows q0 + 16 JQ w .. of a 64 JQ-row tile (JQ 16-row fragments per wave).
// Stage = K image (read both ways) + V image: 16 KB per 64 head-dim columns, 4 DMAs per lane per tile and sub-image.
template <int ND, int JQ>
struct DqCfg {
  static constexpr int NST = ND == 1 ? 3 : 2;
  static constexpr int STAGE = 2 * ND * IMG;
  static constexpr int OCC = ND == 1 ? (JQ == 1 ? 3 : 2) : 2;
};
template <int ND, int JQ>
__global__ __launch_bounds__(256, (DqCfg<ND, JQ>::OCC)) void attn_bwd_dq_kernel(AttnArgs p) {
  constexpr int D = 64 * ND, STG = DqCfg<ND, JQ>::STAGE, NST = DqCfg<ND, JQ>::NST, QT = 64 * JQ, WR = 16 * JQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.M, ld = p.ldq;
  const BlockItem bi = block_item(blockIdx.x, (M + QT - 1) / QT, p.nH, p.nKV);
  if (!bi.valid) return;
  const int h = bi.h, kvh = bi.kvh;
  const int q0 = (p.perm ? p.perm[bi.slot] : bi.slot) * QT;
  if (p.prio && p.perm) set_rank_prio(bi.slot, (M + QT - 1) / QT);
  const int wv = __builtin_amdgcn_readfirstlane(wave);  // SGPR: the wave-uniform tile tests below become scalar branches
  const int qw0 = q0 + wv * WR;
  const bf16_t* Qb = p.qkv + h * D;
  const bf16_t* Kb = p.qkv + (p.nH + kvh) * D;
  const bf16_t* Vb = p.qkv + (p.nH + p.nKV + kvh) * D;
  const float c2 = p.scale * 1.44269504088896340736f;
  const uint32_t lds0 = lds_addr(smem);

  const int kt_begin = p.seg_start[q0 < M ? q0 : M - 1] / 64;
  const int kt_end = (min(q0 + QT - 1, M - 1)) / 64;
  const int n = kt_end - kt_begin + 1;
  TileOff off;
  off.init(ld, tid);
  FragOff fo;
  fo.init(l15, g);
  const size_t tstep = (size_t)64 * ld;
  const bf16_t* kp = Kb + (size_t)kt_begin * tstep;
  const bf16_t* vp = Vb + (size_t)kt_begin * tstep;
  int irow = kt_begin * 64;
  const uint32_t wdst = lds0 + (uint32_t)wv * 1024u;
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const uint32_t st = wdst + (uint32_t)(stage * STG);
    if (irow + 64 <= M) {  // ONE wave-uniform branch per tile
#pragma unroll
      for (int dh = 0; dh < ND; ++dh) {
        dma_tile64<true>(kp + dh * 64, off, 63, st + dh * IMG);
        dma_tile64<true>(vp + dh * 64, off, 63, st + (ND + dh) * IMG);
      }
    } else {
      const int mr = M - 1 - irow;
#pragma unroll
      for (int dh = 0; dh < ND; ++dh) {
        dma_tile64<false>(kp + dh * 64, off, mr, st + dh * IMG);
        dma_tile64<false>(vp + dh * 64, off, mr, st + (ND + dh) * IMG);
      }
    }
    kp += tstep; vp += tstep; irow += 64;
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < n) issue(s);

  int qrow[JQ], segs[JQ];
  float lse[JQ], nds[JQ];
  uint4 qf[JQ][2 * ND], dof[JQ][2 * ND];
#pragma unroll
  for (int j = 0; j < JQ; ++j) {
    const int q = qw0 + j * 16 + l15;
    const int qc = q < M ? q : M - 1;
    qrow[j] = q;
    segs[j] = p.seg_start[qc];
    lse[j] = -p.lse2[(size_t)h * M + qc];  // kept negated: the score accumulators start there
    float dsm = 0.f;  // D[q] = sum_d dO[q][d] * O[q][d]: each lane owns a quarter of the d's, 4 lanes per row
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds) {
      qf[j][ds] = *reinterpret_cast<const uint4*>(Qb + (size_t)qc * ld + g * 8 + 32 * ds);  // pre-scaled: exp2-domain scores
      dof[j][ds] = *reinterpret_cast<const uint4*>(p.d_o + (size_t)qc * p.nH * D + h * D + g * 8 + 32 * ds);
      uint4 of = *reinterpret_cast<const uint4*>(p.o + (size_t)qc * p.nH * D + h * D + g * 8 + 32 * ds);
      float x[8], y[8];
      unpack_bf16x8(dof[j][ds], x);
      unpack_bf16x8(of, y);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsm += x[e] * y[e];
    }
    dsm += __shfl_xor(dsm, 16, 64);
    dsm += __shfl_xor(dsm, 32, 64);
    nds[j] = -dsm;
    if (g == 0 && q < M) {  // consumed by the dK/dV kernel (launched after this one)
      p.ndsum[(size_t)h * M + q] = -dsm;
      p.nlse[(size_t)h * M + q] = lse[j];
    }
  }
  const int segmax_w = p.seg_start[min(qw0 + WR - 1, M - 1)];
  f32x4_t dq[JQ][4 * ND];
#pragma unroll
  for (int j = 0; j < JQ; ++j)
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) dq[j][fd] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  wait_all_loads_visible();

  int stage = 0, istage = (NST - 1) % NST;
  for (int t = 0; t < n; ++t) {
    if (NST >= 3 && t + 1 < n) wait_vmcnt<4 * ND>();
    else wait_vmcnt<0>();
    __syncthreads();
    if (t + NST - 1 < n) issue(istage);
    istage = istage + 1 == NST ? 0 : istage + 1;
    const uint32_t sb = lds0 + (uint32_t)(stage * STG);
    stage = stage + 1 == NST ? 0 : stage + 1;
    const int key0 = (kt_begin + t) * 64;
    if (key0 > qw0 + WR - 1) continue;
    TileAddr ta;
    ta.set(fo, sb);
    f32x4_t st[4][JQ], dp[4][JQ];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        st[f][j] = (f32x4_t){lse[j], lse[j], lse[j], lse[j]};  // s - lse for free (pre-scaled queries): P = exp2(accumulator)
        dp[f][j] = (f32x4_t){nds[j], nds[j], nds[j], nds[j]};  // dP - D for free: the accumulator starts at -D[q]
      }
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const uint4 kf = ta.D((ds >> 1) * IMG, f, ds & 1);
        const uint4 vf = ta.D((ND + (ds >> 1)) * IMG, f, ds & 1);
#pragma unroll
        for (int j = 0; j < JQ; ++j) {
          st[f][j] = mfma16(kf, qf[j][ds], st[f][j]);
          dp[f][j] = mfma16(vf, dof[j][ds], dp[f][j]);
        }
      }
    uint4 dsb[2][JQ];
    auto soft = [&](auto mk) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(mk)::value;
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        int hi = 0, lo = 0;
        if constexpr (MASK) {
          hi = (qrow[j] < M ? qrow[j] : -(1 << 30)) - key0 - 4 * g;  // rows beyond M see nothing
          lo = segs[j] - key0 - 4 * g;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pe = fast_exp2(st[f][j][r]);
            if constexpr (MASK) {
              const bool ok = (f * 16 + r <= hi) & (f * 16 + r >= lo);
              pe = ok ? pe : 0.f;
            }
            st[f][j][r] = pe * dp[f][j][r];
          }
        dsb[0][j] = pack_pair(st[0][j], st[1][j]);
        dsb[1][j] = pack_pair(st[2][j], st[3][j]);
      }
    };
    uint4 kt0[4 * ND];  // K^T fragments of the first contraction half: requested before the exp arithmetic, landing under it
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) kt0[fd] = ta.T((fd >> 2) * IMG, fd & 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    if ((key0 + 63 > qw0) || (key0 < segmax_w) || (qw0 + WR - 1 >= M)) soft(std::true_type{});
    else soft(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    uint4 kt1[4 * ND];
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) kt1[fd] = ta.T((fd >> 2) * IMG, fd & 3, 1);
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
      for (int j = 0; j < JQ; ++j) dq[j][fd] = mfma16(kt0[fd], dsb[0][j], dq[j][fd]);
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
      for (int j = 0; j < JQ; ++j) dq[j][fd] = mfma16(kt1[fd], dsb[1][j], dq[j][fd]);
  }
#pragma unroll
  for (int j = 0; j < JQ; ++j) {
    const int q = qrow[j];
    if (q < M) {
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd)
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[j][fd][r] *= p.scale;
      if (p.rope_cs) {  // transpose rotation: d(pre-RoPE q); fragments fd and fd + 2ND hold d and d + D/2
#pragma unroll
        for (int fd = 0; fd < 2 * ND; ++fd) {
          const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cs + (size_t)q * (D / 2) + fd * 16 + g * 4);
          const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sn + (size_t)q * (D / 2) + fd * 16 + g * 4);
          const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a = dq[j][fd][r], b = dq[j][fd + 2 * ND][r];
            dq[j][fd][r] = a * cc[r] + b * ss[r];
            dq[j][fd + 2 * ND][r] = b * cc[r] - a * ss[r];
          }
        }
      }
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) {
        uint2 o;
        o.x = pack_bf16x2(dq[j][fd][0], dq[j][fd][1]);
        o.y = pack_bf16x2(dq[j][fd][2], dq[j][fd][3]);
        *reinterpret_cast<uint2*>(p.dqkv + (size_t)q * ld + h * D + fd * 16 + g * 4) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// dK / dV. One block = (64 KW-key tile, KV head, query-range chunk); wave w owns keys k0 + 16 KW w .. (KW 16-key
// fragments, K and V rows in registers for the whole block). The block walks the G query heads of the KV group and, per
// head, the 64-row query tiles of its chunk: Q and dO tiles stream through the ring (one image each, read both ways), dK
// and dV accumulate over ALL of it in registers. Stage = Q image + dO image (16 KB per 64 head-dim columns) + -lse2 /
// -D / seg_start of the 64 query rows (3 x 256 B, by 4-byte LDS-DMA): 5 DMAs per lane per tile at head_dim 64.
template <int ND, int KW>
struct DkvCfg {
  static constexpr int NST = ND == 1 ? 3 : 2;
  static constexpr int STAGE = 2 * ND * IMG + 1024;
  static constexpr int OCC = ND == 1 ? (KW == 1 ? 3 : 2) : 2;
};
struct DkvRange { int qa, nq, nch; };
// query tiles (64 rows) that can see the key tile [k0, k0 + KT), cut into at most `nchmax` chunks of equal length
SLAM_DEVICE DkvRange dkv_range(const int* seg_end, int M, int k0, int KT, int nchmax, int c) {
  const int qt_begin = k0 / 64;
  const int qt_end = (seg_end[min(k0 + KT - 1, M - 1)] - 1) / 64;
  const int n = qt_end - qt_begin + 1;
  const int cs = (n + nchmax - 1) / nchmax;
  DkvRange r;
  r.nch = (n + cs - 1) / cs;
  r.qa = qt_begin + c * cs;
  r.nq = min(cs, qt_end - r.qa + 1);
  return r;
}
template <int ND, int KW>
__global__ __launch_bounds__(256, (DkvCfg<ND, KW>::OCC)) void attn_bwd_dkv_kernel(AttnArgs p) {
  constexpr int D = 64 * ND, STG = DkvCfg<ND, KW>::STAGE, NST = DkvCfg<ND, KW>::NST, KT = 64 * KW, WK = 16 * KW;
  constexpr int NDMA = 4 * ND + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.M, ld = p.ldq, ldo = p.nH * D;
  const int G = p.nH / p.nKV;
  const int kvh = blockIdx.x % p.nKV;
  const int item = p.perm[blockIdx.x / p.nKV];
  if (item < 0) return;  // the item list is sorted by work: invalid (key tile, chunk) candidates sit at its end
  if (p.prio) set_rank_prio(blockIdx.x / p.nKV, ((M + KT - 1) / KT) * NCH_MAX);
  const int chunk = item & (NCH_MAX - 1), k0 = (item >> 2) * KT;
  const DkvRange rg = dkv_range(p.seg_end, M, k0, KT, p.nch, chunk);
  const int nq = rg.nq, qa = rg.qa;
  const int n = G * nq;
  const bf16_t* Kb = p.qkv + (p.nH + kvh) * D;
  const bf16_t* Vb = p.qkv + (p.nH + p.nKV + kvh) * D;
  const float c2 = p.scale * 1.44269504088896340736f;
  const uint32_t lds0 = lds_addr(smem);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  TileOff qoff, ooff;
  qoff.init(ld, tid); ooff.init(ldo, tid);
  FragOff fo;
  fo.init(l15, g);

  // next tile to issue: running pointers of the (head, query tile) walk, all wave-uniform
  const size_t qstep = (size_t)64 * ld, ostep = (size_t)64 * ldo;
  const bf16_t* qp = p.qkv + (size_t)(kvh * G) * D + (size_t)qa * qstep;
  const bf16_t* dop = p.d_o + (size_t)(kvh * G) * D + (size_t)qa * ostep;
  int i_tq = 0, i_head = kvh * G, istage = 0, irow = qa * 64;
  const uint32_t wdst = lds0 + (uint32_t)wv * 1024u, sdst = lds0 + (uint32_t)(2 * ND * IMG) + (uint32_t)wv * 256u;
  auto issue_next = [&]() __attribute__((always_inline)) {
    const uint32_t st = wdst + (uint32_t)(istage * STG);
    const int mr = irow + 64 <= M ? 63 : M - 1 - irow;
    if (mr >= 63) {  // ONE wave-uniform branch per tile
#pragma unroll
      for (int dh = 0; dh < ND; ++dh) {
        dma_tile64<true>(qp + dh * 64, qoff, 63, st + dh * IMG);
        dma_tile64<true>(dop + dh * 64, ooff, 63, st + (ND + dh) * IMG);
      }
    } else {
#pragma unroll
      for (int dh = 0; dh < ND; ++dh) {
        dma_tile64<false>(qp + dh * 64, qoff, mr, st + dh * IMG);
        dma_tile64<false>(dop + dh * 64, ooff, mr, st + (ND + dh) * IMG);
      }
    }
    // per-row scalars: wave 0 -> lse2, 1 -> -D, 2 -> seg_start, 3 -> spare slot (keeps the DMA count uniform)
    // (the three sources are chosen from loop-invariant bases: a select between loop-carried pointers becomes a lookup
    //  table in scratch memory, and scratch traffic would also break the counted vmcnt waits)
    const size_t hrow = (size_t)i_head * M + irow;
    const void* src = wv == 0 ? (const void*)(p.nlse + hrow) : wv == 1 ? (const void*)(p.ndsum + hrow) : (const void*)(p.seg_start + irow);
    glds4_m0(src, (uint32_t)(lane < mr ? lane : mr) * 4u, sdst + (uint32_t)(istage * STG));
    qp += qstep; dop += ostep; irow += 64;
    if (++i_tq == nq) {  // next head of the group: back to the chunk's first query tile
      i_tq = 0;
      ++i_head;
      qp += (ptrdiff_t)D - (ptrdiff_t)nq * (ptrdiff_t)qstep; dop += (ptrdiff_t)D - (ptrdiff_t)nq * (ptrdiff_t)ostep;
      irow -= nq * 64;
    }
    istage = istage + 1 == NST ? 0 : istage + 1;
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < n) issue_next();

  int key[KW];
  uint4 kf[KW][2 * ND], vf[KW][2 * ND];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    key[i] = k0 + wv * WK + i * 16 + l15;
    const int kc = key[i] < M ? key[i] : M - 1;
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds) {
      kf[i][ds] = *reinterpret_cast<const uint4*>(Kb + (size_t)kc * ld + g * 8 + 32 * ds);
      vf[i][ds] = *reinterpret_cast<const uint4*>(Vb + (size_t)kc * ld + g * 8 + 32 * ds);
    }
  }
  f32x4_t dk[KW][4 * ND], dv[KW][4 * ND];
#pragma unroll
  for (int i = 0; i < KW; ++i)
#pragma unroll
    for (int fd = 0; fd < 4 * ND; ++fd) { dk[i][fd] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[i][fd] = dk[i][fd]; }
  wait_all_loads_visible();

  const int kw0 = k0 + wv * WK;  // SGPR (wv): the tile tests below are scalar branches
  int stage = 0, tq = 0;
  for (int t = 0; t < n; ++t) {
    if (NST >= 3 && t + 1 < n) wait_vmcnt<NDMA>();
    else wait_vmcnt<0>();
    __syncthreads();
    if (t + NST - 1 < n) issue_next();
    const uint32_t sb = lds0 + (uint32_t)(stage * STG);
    stage = stage + 1 == NST ? 0 : stage + 1;
    const int qbase = (qa + tq) * 64;
    tq = tq + 1 == nq ? 0 : tq + 1;
    if (qbase + 63 < kw0) continue;  // no query of the tile can see this wave's keys
    TileAddr ta;
    ta.set(fo, sb);
    uint32_t sca = sb + (uint32_t)(2 * ND * IMG + g * 16);  // per-row scalars of rows 16 jq + 4g .. +3
    asm volatile("" : "+v"(sca));
    f32x4_t s[4][KW], dp[4][KW];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const f32x4_t nl4 = *(__attribute__((address_space(3))) const f32x4_t*)(lds_p(sca) + jq * 64);
      const f32x4_t nd4 = *(__attribute__((address_space(3))) const f32x4_t*)(lds_p(sca) + (256 + jq * 64));
#pragma unroll
      for (int i = 0; i < KW; ++i) { s[jq][i] = nl4; dp[jq][i] = nd4; }  // s - lse and dP - D for free: the accumulators start there
    }
#pragma unroll
    for (int ds = 0; ds < 2 * ND; ++ds)
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const uint4 qfr = ta.D((ds >> 1) * IMG, jq, ds & 1);
        const uint4 dofr = ta.D((ND + (ds >> 1)) * IMG, jq, ds & 1);
#pragma unroll
        for (int i = 0; i < KW; ++i) {
          s[jq][i] = mfma16(qfr, kf[i][ds], s[jq][i]);
          dp[jq][i] = mfma16(dofr, vf[i][ds], dp[jq][i]);
        }
      }
    // lane holds (q = qbase + jq*16 + 4g + r, key[i]); mask only on diagonal / segment-boundary / tail tiles
    uint4 pb[2][KW], dsb[2][KW];
    auto soft = [&](auto mk) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(mk)::value;
      int lo[KW], mh = 0;
      if constexpr (MASK) {
        mh = M - qbase - 4 * g;  // row 16 jq + r is a real query iff 16 jq + r < mh
#pragma unroll
        for (int i = 0; i < KW; ++i) lo[i] = key[i] - qbase - 4 * g;  // ... and sees key[i] iff 16 jq + r >= lo[i] (and the segment test)
      }
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        int sv[4] = {0, 0, 0, 0};
        if constexpr (MASK) {
          typedef __attribute__((ext_vector_type(4))) int i32x4_t;
          const i32x4_t s4 = *(__attribute__((address_space(3))) const i32x4_t*)(lds_p(sca) + (512 + jq * 64));
          sv[0] = s4[0]; sv[1] = s4[1]; sv[2] = s4[2]; sv[3] = s4[3];
        }
#pragma unroll
        for (int i = 0; i < KW; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pe = fast_exp2(s[jq][i][r]);
            if constexpr (MASK) {
              const bool ok = (jq * 16 + r >= lo[i]) & (jq * 16 + r < mh) & (key[i] >= sv[r]);
              pe = ok ? pe : 0.f;
            }
            s[jq][i][r] = pe;
            dp[jq][i][r] = pe * dp[jq][i][r];
          }
      }
#pragma unroll
      for (int i = 0; i < KW; ++i) {
        pb[0][i] = pack_pair(s[0][i], s[1][i]);
        pb[1][i] = pack_pair(s[2][i], s[3][i]);
        dsb[0][i] = pack_pair(dp[0][i], dp[1][i]);
        dsb[1][i] = pack_pair(dp[2][i], dp[3][i]);
      }
    };
    // latest segment start among the tile's query rows: row 63's (rows beyond M repeat row M-1)
    const int segmax = *(__attribute__((address_space(3))) const int*)(lds_p(sb + (uint32_t)(2 * ND * IMG + 512 + 63 * 4)));
    constexpr bool PF = KW == 1;  // (32 keys per wave: no registers left for it)
    uint4 dot0[4 * ND];  // dO^T fragments of the first contraction half: requested before the exp arithmetic
    if constexpr (PF) {
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) dot0[fd] = ta.T((ND + (fd >> 2)) * IMG, fd & 3, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((kw0 + WK - 1 > qbase) || (kw0 < segmax) || (qbase + 63 >= M)) soft(std::true_type{});
    else soft(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) {
        const uint4 dot = (PF && t2 == 0) ? dot0[fd] : ta.T((ND + (fd >> 2)) * IMG, fd & 3, t2);
        const uint4 qt = ta.T((fd >> 2) * IMG, fd & 3, t2);
#pragma unroll
        for (int i = 0; i < KW; ++i) {
          dv[i][fd] = mfma16(dot, pb[t2][i], dv[i][fd]);
          dk[i][fd] = mfma16(qt, dsb[t2][i], dk[i][fd]);
        }
      }
  }
#pragma unroll
  for (int i = 0; i < KW; ++i)
    if (key[i] < M) {
      float* dkp = p.dkv_part + ((((size_t)chunk * 2 + 0) * p.nKV + kvh) * M + key[i]) * D;
      float* dvp = p.dkv_part + ((((size_t)chunk * 2 + 1) * p.nKV + kvh) * M + key[i]) * D;
#pragma unroll
      for (int fd = 0; fd < 4 * ND; ++fd) {
        *reinterpret_cast<float4*>(dkp + fd * 16 + g * 4) =
            make_float4(dk[i][fd][0] * LN2, dk[i][fd][1] * LN2, dk[i][fd][2] * LN2, dk[i][fd][3] * LN2);  // Q tiles are pre-scaled: scale / (scale log2 e)
        *reinterpret_cast<float4*>(dvp + fd * 16 + g * 4) =
            make_float4(dv[i][fd][0], dv[i][fd][1], dv[i][fd][2], dv[i][fd][3]);
      }
    }
}

// dqkv[m][K head kvh / V head kvh] = bf16( sum over the key tile's chunks, in chunk order ); a thread owns
// d = 4c..4c+3 and its rotate-half partner d + D/2, so dK can be rotated back in the same pass.
template <int ND>
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnArgs p, int KT) {
  constexpr int D = 64 * ND, HALF = D / 2, CPR = D / 8;  // CPR threads per (m, kv head)
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // (which, m, kvh, c < CPR)
  size_t total = (size_t)2 * p.M * p.nKV * CPR;
  if (idx >= total) return;
  int c = idx % CPR;
  size_t r = idx / CPR;
  int kvh = r % p.nKV; r /= p.nKV;
  int m = r % p.M;
  int which = (int)(r / p.M);
  const int nch = dkv_range(p.seg_end, p.M, (m / KT) * KT, KT, p.nch, 0).nch;
  float4 a = make_float4(0, 0, 0, 0), b = a;
  for (int i = 0; i < nch; ++i) {
    const float* src = p.dkv_part + ((((size_t)i * 2 + which) * p.nKV + kvh) * p.M + m) * D + c * 4;
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
// plan = [ fwd perm (128-row q tiles) | dq perm (qt-row q tiles) | dkv items ((key tile << 2) | chunk, -1 = none) ].
__global__ __launch_bounds__(1024) void attn_plan_kernel(const int* __restrict__ seg_s, const int* __restrict__ seg_e, int M, int qt,
                                                        int kt, int nchmax, int* __restrict__ plan) {
  // one block: phase 1 writes every candidate's work into LDS, phase 2 ranks it inside its own list
  extern __shared__ int wk[];
  const int nf = (M + 127) / 128, nq = (M + qt - 1) / qt, nk = (M + kt - 1) / kt;
  const int total = seg_e ? nf + nq + nk * NCH_MAX : nf;  // seg_e == NULL: forward order only
  for (int t = threadIdx.x; t < total; t += blockDim.x) {
    int w;
    if (t < nf) { int q0 = t * 128; w = min(q0 + 127, M - 1) / 64 - seg_s[q0] / 64 + 1; }
    else if (t < nf + nq) { int q0 = (t - nf) * qt; w = min(q0 + qt - 1, M - 1) / 64 - seg_s[q0] / 64 + 1; }
    else {
      const int j = t - nf - nq, c = j & (NCH_MAX - 1);
      const DkvRange r = dkv_range(seg_e, M, (j >> 2) * kt, kt, nchmax, c);
      w = c < r.nch ? r.nq : -1;
    }
    wk[t] = w;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < total; t += blockDim.x) {
    int base, n;
    if (t < nf) { base = 0; n = nf; }
    else if (t < nf + nq) { base = nf; n = nq; }
    else { base = nf + nq; n = nk * NCH_MAX; }
    const int i = t - base, w = wk[t];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const int wj = wk[base + j];
      rank += (wj > w) || (wj == w && j < i);
    }
    plan[base + rank] = (base == nf + nq && w <= 0) ? -1 : i;
  }
}

}  // namespace

namespace slam {

static AttnTune g_attn_tune = {1, 1, 4, 0};
AttnTune attn_default_tune() { return g_attn_tune; }
void attn_set_default_tune(AttnTune t) { g_attn_tune = t; }
static AttnTune clamp_tune(AttnTune t, int head_dim) {
  t.jq = t.jq == 2 && head_dim == 64 ? 2 : 1;
  t.kw = 1;  // (32 keys per wave was built and measured in round 3: 250 registers, two waves per SIMD, never faster - removed)
  t.nch = t.nch < 1 ? 1 : t.nch > NCH_MAX ? NCH_MAX : t.nch;
  return t;
}

size_t attn_plan_ints(int M) { return (size_t)((M + 127) / 128 + (M + 63) / 64 + NCH_MAX * ((M + 63) / 64)); }
int attn_plan(const int* seg_start, const int* seg_end, int M, int head_dim, AttnTune tune, int* plan, hipStream_t st) {
  tune = clamp_tune(tune, head_dim);
  const int qt = 64 * tune.jq, kt = 64 * tune.kw;
  const int total = (M + 127) / 128 + (M + qt - 1) / qt + NCH_MAX * ((M + kt - 1) / kt);
  if ((size_t)total * sizeof(int) > 48 * 1024) return -1;  // M beyond ~260k tokens per micro-batch: the work list no longer fits one block's LDS
  attn_plan_kernel<<<1, 1024, (size_t)total * sizeof(int), st>>>(seg_start, seg_end, M, qt, kt, tune.nch, plan);
  return (int)hipGetLastError();
}

template <typename K>
static int set_lds(K kernel, int bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int ND>
static int attn_fwd_nd(AttnArgs a, hipStream_t st) {
  static bool done = false;
  constexpr int lds = FwdCfg<ND>::NST * FwdCfg<ND>::STAGE;
  if (!done) { if (int e = set_lds(&attn_fwd_kernel<ND>, lds)) return e; done = true; }
  attn_fwd_kernel<ND><<<item_grid((a.M + 127) / 128, a.nH, a.nKV), 256, lds, st>>>(a);
  return (int)hipGetLastError();
}

template <int ND, int JQ>
static int attn_dq_launch(AttnArgs a, hipStream_t st) {
  static bool done = false;
  constexpr int lds = DqCfg<ND, JQ>::NST * DqCfg<ND, JQ>::STAGE;
  if (!done) { if (int e = set_lds(&attn_bwd_dq_kernel<ND, JQ>, lds)) return e; done = true; }
  attn_bwd_dq_kernel<ND, JQ><<<item_grid((a.M + 64 * JQ - 1) / (64 * JQ), a.nH, a.nKV), 256, lds, st>>>(a);
  return (int)hipGetLastError();
}
template <int ND, int KW>
static int attn_dkv_launch(AttnArgs a, hipStream_t st) {
  static bool done = false;
  constexpr int lds = DkvCfg<ND, KW>::NST * DkvCfg<ND, KW>::STAGE;
  if (!done) { if (int e = set_lds(&attn_bwd_dkv_kernel<ND, KW>, lds)) return e; done = true; }
  const int nk = (a.M + 64 * KW - 1) / (64 * KW);
  attn_bwd_dkv_kernel<ND, KW><<<nk * NCH_MAX * a.nKV, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}

int attn_fwd(const bf16_t* qkv, bf16_t* o, float* lse2, const int* seg_start, const int* plan, AttnTune tune, int M, int nH,
             int nKV, int head_dim, hipStream_t st) {
  if ((head_dim != 64 && head_dim != 128) || nH % nKV) return -1;
  AttnArgs a{};
  a.qkv = qkv; a.o = o; a.lse2 = lse2; a.seg_start = seg_start; a.perm = plan; a.prio = tune.prio;
  a.M = M; a.nH = nH; a.nKV = nKV; a.ldq = (nH + 2 * nKV) * head_dim; a.scale = 1.0f / sqrtf((float)head_dim);
  return head_dim == 64 ? attn_fwd_nd<1>(a, st) : attn_fwd_nd<2>(a, st);
}

size_t attn_bwd_workspace_bytes(int M, int nKV, int head_dim) {
  return (size_t)NCH_MAX * 2 * nKV * M * head_dim * sizeof(float);
}

int attn_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse2, float* ndsum, float* nlse,
             bf16_t* dqkv, float* dkv_part, const int* seg_start, const int* seg_end, const int* plan, AttnTune tune,
             const float* rope_cs, const float* rope_sn, int M, int nH, int nKV, int head_dim, hipStream_t st) {
  if ((head_dim != 64 && head_dim != 128) || nH % nKV || !plan) return -1;
  tune = clamp_tune(tune, head_dim);
  AttnArgs a{};
  a.qkv = qkv; a.o = const_cast<bf16_t*>(o); a.d_o = d_o; a.dqkv = dqkv;
  a.lse2 = const_cast<float*>(lse2); a.ndsum = ndsum; a.nlse = nlse; a.dkv_part = dkv_part;
  a.seg_start = seg_start; a.seg_end = seg_end;
  a.rope_cs = rope_cs; a.rope_sn = rope_sn;
  a.M = M; a.nH = nH; a.nKV = nKV; a.ldq = (nH + 2 * nKV) * head_dim; a.scale = 1.0f / sqrtf((float)head_dim);
  a.nch = tune.nch; a.prio = tune.prio;
  const int nf = (M + 127) / 128, qt = 64 * tune.jq, kt = 64 * tune.kw;
  a.perm = plan + nf;
  int e;
  if (head_dim == 128) e = attn_dq_launch<2, 1>(a, st);
  else e = tune.jq == 2 ? attn_dq_launch<1, 2>(a, st) : attn_dq_launch<1, 1>(a, st);
  if (e) return e;
  a.perm = plan + nf + (M + qt - 1) / qt;
  if (head_dim == 128) e = attn_dkv_launch<2, 1>(a, st);
  else e = attn_dkv_launch<1, 1>(a, st);
  if (e) return e;
  const size_t total = (size_t)2 * M * nKV * (head_dim / 8);
  if (head_dim == 128) attn_dkv_reduce_kernel<2><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, kt);
  else attn_dkv_reduce_kernel<1><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, kt);
  return (int)hipGetLastError();
}

}  // namespace slam
