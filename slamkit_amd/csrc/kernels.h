// Internal launcher declarations shared by the engine translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef uint16_t bf16_t;

namespace slam {

// gemm.hip
// Kernel-selection / planning knobs of the GEMM dispatch (DESIGN.md section 4 has the measurement behind each default). One
// instance per engine; an engine entry point makes its instance current for the duration of the call (GemmTuneScope).
struct GemmTune {
  int glds = 1;                 // 1 = LDS-DMA staging where the shape allows, 0 = register staging everywhere (parity tests)
  int tn_dma = 1;
  int group_rows = 3, group_rows_256 = 4;   // row-tile groups of the L2-aware block order (128 / 256-row tiles)
  int nt_store = 0;
  int g256 = 1;                 // 256 x 256 NT kernel: 1 = when its fill criterion holds, 0 = never, 2 = whenever the shape allows
  int g256_dswiglu = 1, g256_persist = 1;
  // late start (10-ns ticks) of the persistent blocks that have a tile of slack (plain / SwiGLU forward; SwiGLU backward):
  // their store bursts fall under the other blocks' K loops. Same box, interleaved: 24.88 / 24.95 ms vs 25.05 / 25.17 / 25.39
  int g256_stagger = 1200, g256_stagger_dswiglu = 1200;
  int group_cols_256 = 0;       // > 0: 256-row tile groups are group_rows_256 x group_cols_256 tiles (0 = all columns of a band)
  int g256_persist_cus = 0;     // > 0: blocks of the persistent 256 x 256 grids (multiple of 8; 0 = one per CU): CUs left to RCCL under data parallelism
  int g256_cohorts = 0;         // > 1: start offsets for every persistent block, cohort c of the XCD's slots c * stagger late (probe)
  int shared = 0;               // the launches of this call share the GPU with the engine's wgrad stream (set inside slam_backward)
  int nt224 = 1, nt224_min_k = 2048;
  int tn_splits_override = 0, tn_balanced = 1, bal_bg_max_split = 4;
  int tn224 = 1, tn224_min_m = 16384, tn224_max_split = 16, tn224_bg_min_m = 4096, tn224_bg_max_split = 1;
  // round 5: main loops on v_mfma_f32_32x32x16_bf16 (1) or on the 16x16x32 form (0) in the kernels that have both
  // (128 x 128 NT DMA kernel, 256 x 256 kernels); the 32x32 form sums a 64-deep K-tile in four steps of 16, the 16x16 form
  // in two of 32: results differ in the last bits between the two settings, not between tile shapes under one setting.
  // Measured and OFF (profiles/r5_power_or_stall.md): the 32x32 form needs 3 % fewer cycles and runs at a 7 % lower clock
  // under the board's power limit; Slam-358M step 330.7 k (8 waves) / 323.4 k (4 waves) vs 338.8 k tokens/s
  int mf32 = 0;
  int g256_w4 = 0;              // 256 x 256 tiles on the persistent four-wave kernel (128 x 128 per wave; needs mf32)
  // round 6: the persistent 256 x 256 kernel with its two wave rows as LOADERS (every LDS-DMA, every counted wait, results handed
  // over through LDS) and STORERS (every output store, no vmcnt wait in the K loop): the store drain of a tile runs under the
  // next tile's K loop instead of in front of it (loads and stores share one in-order vmcnt per wave)
  int g256_roles = 0;
  // round 6: the persistent 256 x 256 SwiGLU-backward epilogue issues the gate|up loads of a 64-row quadrant together and the
  // second quadrant's before the first one's stores (two load round trips per tile instead of eight, none behind a store)
  int g256_batch_loads = 1;
};
GemmTune* gemm_default_tune();
GemmTune* gemm_use_tune(GemmTune* t);  // install t (NULL = process default) for this thread; returns the previous one
int gemm_tune_set(GemmTune* t, const char* key, long value);  // "gemm_*" option keys of slam_set_option; 1 = set, 0 = unknown key, -1 = out of range
struct GemmTuneScope {
  GemmTune* old;
  explicit GemmTuneScope(GemmTune* t) : old(gemm_use_tune(t)) {}
  ~GemmTuneScope() { gemm_use_tune(old); }
};
int gemm_nt(const bf16_t* X, const bf16_t* W, bf16_t* Y, const bf16_t* bias, const bf16_t* resid, int M, int N,
            int K, hipStream_t st);
// same, and additionally act[M][N/2] = silu(gate) * up for W rows laid out in 32-row gate/up blocks
int gemm_nt_swiglu(const bf16_t* X, const bf16_t* W, bf16_t* Y, bf16_t* act, int M, int N, int K, hipStream_t st);
// rotate-half RoPE on the first rope_heads 64-column heads; the first q_heads of them with the (pre-scaled) csq / snq tables
int gemm_nt_rope(const bf16_t* X, const bf16_t* W, bf16_t* Y, const bf16_t* bias, const float* cs, const float* sn,
                 const float* csq, const float* snq, int q_heads, int rope_heads, int M, int N, int K, hipStream_t st);
int gemm_nt_dswiglu(const bf16_t* dY, const bf16_t* Wt, bf16_t* gu, int M, int N, int K, hipStream_t st);
int gemm_nn(const bf16_t* dY, const bf16_t* W, bf16_t* dX, const bf16_t* resid, int M, int N, int K,
            hipStream_t st);
int gemm_tn_splits(int M, int N, int K);
// How a launch delivers the FINAL values of a gradient tensor - the values the clip and the optimizer consume (round 6).
// The last backward of an optimizer step may keep them in bf16 only, like the reference does (its parameters, hence its
// gradients, are bf16: /root/reference config/model/slam.yaml:9), and emits the sum of squares of what it stored from the
// same registers: no fp32 gradient store, no norm pass over the buffer.
struct GradSink {
  int img_only = 0;        // 1: final values are stored to the bf16 image ONLY (partial sums keep using dW / slabs in fp32)
  float* sumsq = nullptr;  // one partial sum of squares per block, of the final values AS KEPT (the rounded ones when img_only),
                           // added in a fixed order inside the block: block b of the GEMM kernel -> sumsq[b], block c
                           // (y-major) of its reduce kernel -> sumsq[gemm blocks + c]. Blocks without a final store write
                           // nothing: the caller clears the slots first.
  int cap = 0;             // slots behind sumsq; a plan that needs more fails with -3
  int used = 0;            // out: slots this launch owns
};
// upper bound of GradSink::used for dW[N][K] under every plan gemm_tn can choose
size_t gemm_tn_sumsq_slots(int N, int K);
size_t gemm_tn_workspace_bytes(int M, int N, int K);
// background = 1: the launch shares the GPU with other streams (the engine's wgrad side stream): plans for CU-time per
// flop instead of chip fill (no K-splitting on the 256 x 224 kernel)
// ws_bytes: capacity of `ws` (from gemm_tn_workspace_bytes(Mmax, N, K) with Mmax >= M): a plan that does not fit returns -3
// img (nullable): bf16 image of dW with the same indexing - every FINAL value of dW (unsplit tile epilogues, slab reduces) is
// also stored there rounded to nearest even: the communication image of a bf16 gradient exchange, without a conversion pass
int gemm_tn(const bf16_t* dY, const bf16_t* X, float* dW, int accumulate, int M, int N, int K, int ldy, int ldx,
            float* ws, size_t ws_bytes, hipStream_t st, int background = 0, bf16_t* img = nullptr, GradSink* sink = nullptr);

// attention.hip
// launch-shape choices of the backward kernels (engine-owned, "attn_jq" / "attn_kw" / "attn_nch" options):
// jq / kw = 16-row fragments per wave in dQ / dK-dV (1 or 2; head_dim 64 only), nch = query-range chunks per key tile (1..4)
struct AttnTune { int jq, kw, nch, prio; };  // prio: s_setprio by LPT rank (0 = off)
AttnTune attn_default_tune();
void attn_set_default_tune(AttnTune t);
size_t attn_plan_ints(int M);
int attn_plan(const int* seg_start, const int* seg_end, int M, int head_dim, AttnTune tune, int* plan, hipStream_t st);
int attn_fwd(const bf16_t* qkv, bf16_t* o, float* lse2, const int* seg_start, const int* plan, AttnTune tune, int M, int nH,
             int nKV, int head_dim, hipStream_t st);
size_t attn_bwd_workspace_bytes(int M, int nKV, int head_dim);
// rope_cs / rope_sn (nullable): fp32 [M][head_dim/2] tables; when given, dq and dk are written already
// rotated back (transpose rotation), i.e. as gradients of the pre-RoPE projections. plan (required) must come from
// attn_plan with the same tune.
// ndsum, nlse: fp32 [nH * M] scratch each (-rowsum(dO*O) and -lse2, written by the dQ kernel for the dK/dV kernel)
int attn_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse2, float* ndsum, float* nlse, bf16_t* dqkv,
             float* dkv_part, const int* seg_start, const int* seg_end, const int* plan, AttnTune tune,
             const float* rope_cs, const float* rope_sn, int M, int nH, int nKV, int head_dim, hipStream_t st);

// elementwise.hip
int rmsnorm_fwd(const bf16_t* x, const bf16_t* w, bf16_t* y, float* rstd, int M, int H, float eps, hipStream_t st);
int rmsnorm_bwd_blocks(int M);
int rmsnorm_bwd(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const float* rstd, const bf16_t* dres,
                bf16_t* dx, float* dw, int accumulate, float* part, int M, int H, hipStream_t st, bf16_t* dw_img = nullptr,
                GradSink* sink = nullptr);
int colsum_blocks(int M);
int colsum_bf16(const bf16_t* X, int ld, int M, int N, float* out, int accumulate, float* part, hipStream_t st);
// csq / snq (nullable): the same tables times qscale - the QUERY heads are rotated with these, so that q is stored
// pre-scaled by head_dim^-0.5 * log2(e) (one rounding): the attention kernels' scores come out in the exp2 domain
int rope_table(const int64_t* pos, int M, int T, int head_dim, float theta, float* cs, float* sn, float* csq, float* snq,
               float qscale, hipStream_t st);
int rope_apply(bf16_t* qkv, int ld, int M, int nrot_heads, int head_dim, const float* cs, const float* sn, int backward,
               hipStream_t st, int q_heads = 0, float q_scale = 1.f);
int swiglu_fwd(const bf16_t* gu, bf16_t* act, int M, int I, int blk, hipStream_t st);
int swiglu_bwd(bf16_t* gu, const bf16_t* dact, int M, int I, int blk, hipStream_t st);
int embed_fwd(const int64_t* ids, const bf16_t* E, bf16_t* out, int M, int H, int V, hipStream_t st);
int onehot(const int64_t* ids, bf16_t* oh, int M, int Vp, int V, int pad_id, hipStream_t st);
// gather-side embedding gradient for large vocabularies: dE[ids[m]] += dh[m] in token order (deterministic);
// ws = embed_bwd_workspace_ints(M, Vp) ints
size_t embed_bwd_workspace_ints(int M, int Vp);
int embed_bwd(const int64_t* ids, const bf16_t* dh, float* dE, int M, int H, int Vp, int V, int pad_id, int* ws,
              hipStream_t st);
int cross_entropy(const bf16_t* logits, const int64_t* labels, double num_items, bf16_t* dlogits, float* row_loss,
                  float* denom, float* loss, int B, int T, int Vp, int V, const uint8_t* colmask, hipStream_t st);
int seq_loglik(const float* row_loss, const int64_t* labels, int B, int T, float* ll, float* cnt, hipStream_t st);
int copy_cols(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int M, int ncols, hipStream_t st);
int scale_bf16(bf16_t* x, size_t n, float s, hipStream_t st);
int scale_rows_bf16(bf16_t* x, const float* coef, int M, int T, int ncols, hipStream_t st);
// part: (n + grad_chunk_elems() - 1) / grad_chunk_elems() floats
int grad_norm(const float* g, size_t n, float max_norm, float* part, float* out, hipStream_t st);
int grad_chunk_elems();
int grad_sumsq_chunks(const void* g, int g_bf16, size_t n, size_t off, size_t cnt, float* chunk_sums, hipStream_t st);
int grad_norm_from_chunks(const float* chunk_sums, size_t n_chunks, float max_norm, float* out, hipStream_t st);
// g_bf16 (here and below): the gradients at g are bf16_t (the last backward kept its final values in bf16 only), else float
int adamw(float* p, bf16_t* pb, void* g, int g_bf16, float* m, float* v, size_t n, const float* clip, double lr, double b1,
          double b2, double eps, double wd, int step, int zero_grad, hipStream_t st);
// bf16 parameters and bf16 moments updated in place (fp32 arithmetic per element, no master copy)
int adamw_bf16(bf16_t* p, void* g, int g_bf16, bf16_t* m, bf16_t* v, size_t n, const float* clip, double lr, double b1, double b2,
               double eps, double wd, int step, int zero_grad, hipStream_t st);
int f32_to_bf16(const float* s, bf16_t* d, size_t n, hipStream_t st);
// the same conversion, emitting one GradSink partial (sum of squares of the rounded values) per 8192-element block
int f32_to_bf16_sumsq_slots(size_t n);
int f32_to_bf16_sumsq(const float* s, bf16_t* d, size_t n, float* sumsq, hipStream_t st);
int bf16_to_f32(const bf16_t* s, float* d, size_t n, hipStream_t st);
// AdamW over `batch` same-shaped [R][C] matrices (64-multiples) at a constant stride that ALSO writes the transposed bf16
// image pt[C][R]; mode 0 = fp32 master + fp32 moments, 1 = fp32 master + bf16 moments, 2 = bf16 parameters + bf16 moments.
int adamw_tiles(int mode, float* p, bf16_t* pb, bf16_t* pt, void* g, int g_bf16, void* m, void* v, int R, int C, int batch,
                size_t batch_stride, const float* clip, double lr, double b1, double b2, double eps, double wd, int step, int zero_grad,
                hipStream_t st);
// the same update on `batch` vectors of n elements at a constant stride (no transposed image)
int adamw_strided(int mode, float* p, bf16_t* pb, void* g, int g_bf16, void* m, void* v, size_t n, int batch, size_t stride,
                  const float* clip, double lr, double b1, double b2, double eps, double wd, int step, int zero_grad, hipStream_t st);
int transpose_bf16(const bf16_t* src, bf16_t* dst, int R, int C, int batch, size_t batch_stride, hipStream_t st);
int colsum_finish_many(const float* part, size_t part_stride, int nb, int N, float* out, size_t out_stride, int count,
                       int accumulate, hipStream_t st, bf16_t* img = nullptr,  // img: bf16 image of `out` (same indexing), nullable
                       GradSink* sink = nullptr);

}  // namespace slam
