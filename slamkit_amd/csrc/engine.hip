// C ABI + step orchestration of the Slam engine (see include/slam_engine.h).
// Replaces everything below UnitLM.forward / Trainer.training_step in the reference call stack
// (SURVEY.md §3.1-3.2): Qwen2Model.forward (modeling_qwen2.py:342-402), the tied lm_head (:465),
// compute_loss (unit_lm.py:13-29), autograd backward, clip_grad_norm_ and AdamW.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <dlfcn.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

// RCCL: types only - the library is looked up at run time (slam_comm_*). A box without the RCCL headers still builds the engine
// (the handful of declarations below follow rccl.h / nccl.h; their values are part of the library's stable ABI).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include "../../include/slam_engine.h"
#include "kernels.h"

using namespace slam;

namespace {

constexpr int VPAD_SMALL = 512;  // vocabularies up to 512 (the speech-unit LMs) use the one-wave CE and one-hot wgrad paths
constexpr int GU_BLK = 32;  // Wgu rows / gate|up columns come in blocks of 32 gate + 32 up

struct LayerOff {
  int64_t ln1, wqkv, bqkv, wo, ln2, wgu, wd;
};

struct LayerAct {
  bf16_t *hmid, *x1, *x2, *qkv, *o, *gu, *act;
  float *rstd1, *rstd2, *lse;
};

__global__ void seg_fill_kernel(int* seg_start, int* seg_end, int M, int T) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  int s = (m / T) * T;
  seg_start[m] = s;
  seg_end[m] = s + T;
}

}  // namespace

struct SlamEngine {
  SlamModelDesc d;
  int QKV;  // (nH + 2 nKV) * hd
  int vpad = VPAD_SMALL;  // embedding / logits rows: 512, or vocab rounded up to 256 beyond that
  int64_t n_params;
  int64_t off_embed, off_norm, layer_stride = 0;
  std::vector<LayerOff> lo;
  std::vector<SlamTensorInfo> tensors;
  std::string err;

  bf16_t* params = nullptr;
  bf16_t* params_t = nullptr;  // transposed weight images (optional)
  float* grads = nullptr;

  // workspace
  char* ws = nullptr;
  size_t ws_bytes = 0;
  int64_t max_tokens = 0;
  std::vector<bf16_t*> hs;  // L+1 residual streams
  std::vector<LayerAct> la;
  bf16_t *hf, *logits, *dlogits, *onehot, *dh_a, *dx, *dact, *dqkv, *d_o;
  int* embed_ws = nullptr;
  const uint8_t* logit_mask = nullptr;  // optional [vpad] bytes: non-zero = column excluded from the softmax

  // optimizer overlap ("overlap_adamw"): AdamW + the weight-image refresh run in per-layer chunks on an
  // engine-owned side stream; the next forward waits for chunk l right before layer l, so the HBM-bound update of
  // the later layers runs under the MFMA-bound first layers of the next step
  bool overwrite_next = false;  // "grad_overwrite_next": the next backward stores gradients instead of adding to them
  bf16_t* grad_img = nullptr;   // slam_set_grad_image: the next backward also writes every final gradient value there, as bf16
  // "grad_final_next" (round 6): the next backward is the LAST of its optimizer step. 1: every kernel that stores a final
  // gradient value also emits the sum of squares of its block (GradSink, kernels.h) - slam_grad_norm adds ~60 k partials
  // instead of reading the buffer again. 2: additionally the final values go to the bf16 image ONLY (the reference's own
  // gradient precision, config/model/slam.yaml:9): slam_grad_norm / slam_adamw_* then read the image (2 B/param instead of 4).
  int final_next = 0;
  int gfinal = 0;               // what the last backward did (0: plain - gradients in `grads`, norm from chunk sums)
  bf16_t* g16 = nullptr;        // gfinal == 2: where the gradients are
  float* gn_part = nullptr;     // sum-of-squares partials of the last final backward
  size_t gn_cap = 0, gn_used = 0;
  int norm_partials = 1;        // "grad_norm_partials" = 0: never emit them (slam_grad_norm always takes the chunked pass)
  bool gn_valid = false;        // ... and whether they describe the gradients the clip will see: not when that backward reported
                                // buckets to a callback (data parallel: the norm is that of the EXCHANGED gradients, from chunk sums)
  int overlap_adamw = 0;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_chunk;  // [0] embedding, [1 + l] layer l, [L + 1] final norm
  bool opt_pending = false;

  // "bwd_wgrad_stream": the weight-gradient GEMMs of backward run on a second engine-owned stream. They are off the
  // critical path (nothing in backward reads a weight gradient), so their blocks fill the CU slots the dgrad / attention
  // chain leaves idle: the N = 896 launches occupy 448 of 512 block slots, and a kernel in its HBM-bound epilogue
  // (down-proj dgrad with the fused SwiGLU backward) leaves the MFMA pipes free. Event pairs order every wgrad after the
  // kernel that produces its operands and every buffer re-use on the main stream after the wgrad that reads it.
  AttnTune attn_tune = attn_default_tune();
  GemmTune gemm_tune = *gemm_default_tune();
  int wgrad_stream = 1;  // measured +3.6 % step throughput on Slam-358M (282.2k -> 292.3k tok/s, same box)
  // "bwd_aux_side" (round 6): the launches of backward that nothing on the dgrad chain reads - the bias column sums of d(qkv)
  // and the norm / bias finish kernels - go to the weight-gradient stream as well: the caller's stream IS the critical path
  // (every kernel on it feeds the next), the weight-gradient stream has ~230 us of slack per layer
  int aux_side = 1;
  hipStream_t wside = nullptr;
  // "bwd_wgrad_cus" > 0: the wgrad stream is created with a CU mask of that many CUs (the low bits of the mask: on gfx950
  // bit i is XCD i % 8, shader engine (i / 8) % 4, CU i / 32 - a prefix of 32 k bits is k CUs in every shader engine of every
  // XCD), so that the dgrad / attention / RMSNorm chain on the caller's stream always finds CUs the background GEMMs cannot
  // occupy. hipExtStreamCreateWithCUMask makes a BLOCKING stream: it synchronises implicitly with the NULL stream, so the
  // caller must then run the step on a non-default stream (the trainer and bench.py do when the option is set).
  int wside_cus = 0, wside_cus_applied = 0;
  bool time_gateup = false;        // "time_gateup": timing events around every gate|up projection launch of a forward
  std::vector<hipEvent_t> tg_ev;   // 2 per layer
  // "time_families": timing-event pairs around the launches of every family of slam_forward / slam_backward (slam_family_ms),
  // each on the stream the launch goes to - the in-step duration of a launch between its real neighbours, what a kernel
  // trace shows for it. Two more packets per launch: measurement steps only.
  bool time_families = false;
  std::vector<hipEvent_t> fam_ev;                 // pool, reused step after step
  std::vector<std::pair<int, size_t>> fam_marks;  // (family id, index of the pair's first event)
  hipStream_t bucket_stream = nullptr;  // see slam_bucket_stream
  // engine-side gradient exchange (slam_comm_*): one RCCL communicator, one communication stream, an event pool
  void* comm = nullptr;                 // ncclComm_t
  int comm_world = 0, comm_rank = 0;
  std::vector<hipEvent_t> ag_ev;        // "parameters of this bucket have arrived" (slam_allgather_params_async -> the next forward)
  size_t ag_ev_used = 0;
  hipStream_t comm_stream = nullptr;
  std::vector<hipEvent_t> comm_ev;
  size_t comm_ev_used = 0;
  bf16_t* last_grad_img = nullptr;      // the image the last slam_backward wrote (slam_allreduce_grads_async, bf16 exchange)
  std::vector<hipEvent_t> ev_w;  // per layer (+1 for the head / embedding): 4 main->side, 3 side->main

  ~SlamEngine() {
    if (wside) { (void)hipStreamSynchronize(wside); (void)hipStreamDestroy(wside); }
    for (hipEvent_t e : fam_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_w) (void)hipEventDestroy(e);
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (hipEvent_t e : ev_chunk) (void)hipEventDestroy(e);
    for (hipEvent_t e : pw_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : tg_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : comm_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ag_ev) (void)hipEventDestroy(e);
    if (comm_stream) { (void)hipStreamSynchronize(comm_stream); (void)hipStreamDestroy(comm_stream); }
  }
  // parameter ranges another stream is still writing (sharded optimizer: the bf16 parameter all-gather on the
  // communication stream): the next reader waits for the event right before its first read of the range
  struct ParamWait { int64_t lo, hi; hipEvent_t ev; };
  std::vector<ParamWait> pwaits;
  std::vector<hipEvent_t> pw_ev;   // timing pairs around the waits (exposed all-gather time), reused step after step
  size_t pw_used = 0;
  double pw_acc_ms = 0.0;          // waits whose event pairs were folded into a sum when the pool filled up (not yet reported)
  int64_t pw_untimed = 0;          // waits that could not be bracketed (pool full of pairs still in flight): reported, never silent
  bool params_t_dirty = false;     // ranged optimizer updates leave the transposed weight images stale until backward needs them
  bool time_param_waits = false;   // "time_param_waits": bracket the parameter waits of a forward with timing events (slam_param_wait_ms)
  float* nlse = nullptr;
  float *cosq = nullptr, *sinq = nullptr;  // the query heads' RoPE tables: cos / sin times head_dim^-0.5 * log2(e)
  float *rstdf, *row_loss, *dsum, *dkv_part, *cosb, *sinb, *gemm_ws, *part_ws, *scal;
  float *ln_part, *bias_part;  // per-layer partial slabs: [2L][nb_ln][H], [L][nb_cs][QKV]
  size_t ln_ps = 0, bias_ps = 0;
  size_t gemm_ws_bytes = 0;
  int *seg_s, *seg_e, *attn_plan_buf;

  // last forward
  int B = 0, T = 0;
  bool have_fwd = false, have_loss = false;
  bool fuse_swiglu = false, fuse_dswiglu = true;
  bool fuse_adamw_t = true;  // "fuse_adamw_t": AdamW writes the transposed weight images itself (0: separate transpose pass)
  const int64_t* last_ids = nullptr;
  const int* cur_seg_s = nullptr;
  const int* cur_seg_e = nullptr;

  int fail(int code, const std::string& m) {
    err = m;
    return code;
  }
};

namespace {

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

size_t max_gemm_ws(const SlamEngine* e, int M) {
  const SlamModelDesc& d = e->d;
  size_t w = 0;
  auto upd = [&](int N, int K) { size_t b = gemm_tn_workspace_bytes(M, N, K); if (b > w) w = b; };
  upd(e->QKV, d.hidden);
  upd(d.hidden, d.n_heads * d.head_dim);
  upd(2 * d.intermediate, d.hidden);
  upd(d.hidden, d.intermediate);
  upd(e->vpad, d.hidden);
  return w;
}

size_t carve(SlamEngine* e, char* base, int64_t Mmax) {
  const SlamModelDesc& d = e->d;
  const size_t M = (size_t)Mmax, H = d.hidden, I = d.intermediate, L = d.n_layers;
  Carver c{base};
  e->hs.resize(L + 1);
  e->la.resize(L);
  for (size_t l = 0; l <= L; ++l) e->hs[l] = c.take<bf16_t>(M * H);
  for (size_t l = 0; l < L; ++l) {
    LayerAct& a = e->la[l];
    a.hmid = c.take<bf16_t>(M * H);
    a.x1 = c.take<bf16_t>(M * H);
    a.x2 = c.take<bf16_t>(M * H);
    a.qkv = c.take<bf16_t>(M * e->QKV);
    a.o = c.take<bf16_t>(M * d.n_heads * d.head_dim);
    a.gu = c.take<bf16_t>(M * 2 * I);
    a.act = c.take<bf16_t>(M * I);
    a.rstd1 = c.take<float>(M);
    a.rstd2 = c.take<float>(M);
    a.lse = c.take<float>(M * d.n_heads);
  }
  e->hf = c.take<bf16_t>(M * H);
  e->rstdf = c.take<float>(M);
  const size_t VP = (size_t)e->vpad;
  // ONE [M][Vp] buffer: the loss kernel replaces the logits with d loss / d logits in place (2 x 5 GB -> 5 GB at
  // M = 16384, Vp = 152,320); callers that want the logits get their copy before the loss runs
  e->logits = c.take<bf16_t>(M * VP);
  e->dlogits = e->logits;
  if (e->vpad == VPAD_SMALL) { e->onehot = c.take<bf16_t>(M * VP); e->embed_ws = nullptr; }
  else { e->onehot = nullptr; e->embed_ws = c.take<int>(embed_bwd_workspace_ints((int)M, e->vpad)); }
  e->row_loss = c.take<float>(M);
  e->dh_a = c.take<bf16_t>(M * H);
  e->dx = c.take<bf16_t>(M * H);
  e->dact = c.take<bf16_t>(M * I);
  e->dqkv = c.take<bf16_t>(M * e->QKV);
  e->d_o = c.take<bf16_t>(M * d.n_heads * d.head_dim);
  e->dsum = c.take<float>(M * d.n_heads);
  e->nlse = c.take<float>(M * d.n_heads);
  e->dkv_part = c.take<float>(attn_bwd_workspace_bytes((int)M, d.n_kv_heads, d.head_dim) / sizeof(float));
  e->cosb = c.take<float>(M * (d.head_dim / 2));
  e->sinb = c.take<float>(M * (d.head_dim / 2));
  e->cosq = c.take<float>(M * (d.head_dim / 2));
  e->sinq = c.take<float>(M * (d.head_dim / 2));
  e->seg_s = c.take<int>(M);
  e->seg_e = c.take<int>(M);
  e->attn_plan_buf = c.take<int>(attn_plan_ints((int)M));
  e->gemm_ws_bytes = max_gemm_ws(e, (int)M);
  e->gemm_ws = c.take<float>(e->gemm_ws_bytes / sizeof(float));
  size_t part = (size_t)rmsnorm_bwd_blocks((int)M) * H;
  size_t part2 = (size_t)colsum_blocks((int)M) * e->QKV;
  if (part2 > part) part = part2;
  if (part < 1024) part = 1024;
  const size_t n_chunks = ((size_t)e->n_params + grad_chunk_elems() - 1) / grad_chunk_elems();  // gradient-norm chunk sums
  if (part < n_chunks) part = n_chunks;
  e->part_ws = c.take<float>(part);
  e->ln_ps = (size_t)rmsnorm_bwd_blocks((int)M) * H;
  e->bias_ps = (size_t)colsum_blocks((int)M) * e->QKV;
  e->ln_part = c.take<float>(e->ln_ps * 2 * L);
  e->bias_part = c.take<float>(e->bias_ps * L);
  {  // GradSink slots of one backward: every launch that stores final gradient values, under any plan
    const size_t HD = (size_t)d.n_heads * d.head_dim;
    size_t emb = gemm_tn_sumsq_slots(e->vpad, (int)H);
    const size_t conv = (size_t)f32_to_bf16_sumsq_slots((size_t)e->vpad * H);
    if (conv > emb) emb = conv;
    const size_t per_layer = gemm_tn_sumsq_slots(e->QKV, (int)H) + gemm_tn_sumsq_slots((int)H, (int)HD) +
                             gemm_tn_sumsq_slots((int)(2 * I), (int)H) + gemm_tn_sumsq_slots((int)H, (int)I) +
                             2 * ((H + 15) / 16) + ((size_t)e->QKV + 15) / 16;
    e->gn_cap = emb + L * per_layer + (H + 15) / 16 + 64;
    e->gn_part = c.take<float>(e->gn_cap);
  }
  e->scal = c.take<float>(64);
  return (c.off + 255) & ~(size_t)255;
}

void add_tensor(SlamEngine* e, const std::string& name, int64_t& off, int64_t rows, int64_t cols) {
  SlamTensorInfo t;
  memset(&t, 0, sizeof(t));
  snprintf(t.name, sizeof(t.name), "%s", name.c_str());
  t.offset = off;
  t.rows = rows;
  t.cols = cols;
  e->tensors.push_back(t);
  off += rows * cols;
}

#define CK(expr)                                                                     \
  do {                                                                               \
    int _r = (expr);                                                                 \
    if (_r != 0) {                                                                   \
      char _b[256];                                                                  \
      snprintf(_b, sizeof(_b), "%s failed with %d (%s) at %s:%d", #expr, _r,         \
               _r > 0 ? hipGetErrorString((hipError_t)_r) : "engine", __FILE__, __LINE__); \
      return h->fail(_r, _b);                                                        \
    }                                                                                \
  } while (0)

// Flags of the events that only order the engine's streams among themselves. Without hipEventDisableSystemFence every
// hipEventRecord carries a SYSTEM-scope release - cache write-back and invalidation in the middle of the backward, seven
// times per layer; nothing on the host reads device memory through these events (SLAM_EVENT_SYSTEM_FENCE=1 restores it).
static unsigned sync_event_flags() {
  static const unsigned f = [] {
    const char* e = getenv("SLAM_EVENT_SYSTEM_FENCE");
    return (unsigned)hipEventDisableTiming | ((e && e[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
  }();
  return f;
}

// launch families of the step (slam_family_name); ids are stable within a library build only
enum Fam {
  F_QKV_FWD, F_ATTN_FWD, F_O_FWD, F_GATEUP_FWD, F_DOWN_FWD, F_NORM_FWD, F_HEAD_FWD, F_LOSS,
  F_DOWN_DGRAD, F_GATEUP_DGRAD, F_O_DGRAD, F_QKV_DGRAD, F_ATTN_BWD, F_NORM_BWD, F_HEAD_DGRAD,
  F_WD_WGRAD, F_WGU_WGRAD, F_WO_WGRAD, F_WQKV_WGRAD, F_HEAD_WGRAD, F_EMBED_WGRAD, F_COUNT
};
const char* const kFamName[F_COUNT] = {
    "qkv_fwd", "attn_fwd", "o_fwd", "gateup_fwd", "down_fwd", "norm_fwd", "head_fwd", "loss",
    "down_dgrad_dswiglu", "gateup_dgrad", "o_dgrad", "qkv_dgrad", "attn_bwd", "norm_bwd", "head_dgrad",
    "wd_wgrad", "wgu_wgrad", "wo_wgrad", "wqkv_wgrad", "head_wgrad", "embed_wgrad"};

// first event of a timing pair around a launch of family `fam` on `st`; returns the pair's slot, -1 when timing is off or failed
int fam_begin(SlamEngine* h, int fam, hipStream_t st) {
  if (!h->time_families) return -1;
  const size_t at = h->fam_marks.size() * 2;
  while (h->fam_ev.size() < at + 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    h->fam_ev.push_back(e);
  }
  if (hipEventRecord(h->fam_ev[at], st) != hipSuccess) return -1;
  h->fam_marks.push_back({fam, at});
  return (int)at;
}
void fam_end(SlamEngine* h, int slot, hipStream_t st) {
  if (slot >= 0) (void)hipEventRecord(h->fam_ev[(size_t)slot + 1], st);
}
// CK around a launch (or a short run of launches) of one family on one stream
#define TK(fam, stream_, expr)                      \
  do {                                              \
    const int _slot = fam_begin(h, (fam), (stream_)); \
    CK(expr);                                       \
    fam_end(h, _slot, (stream_));                   \
  } while (0)

int ensure_side(SlamEngine* h) {
  if (h->side) return 0;
  hipError_t e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
  if (e != hipSuccess) return (int)e;
  e = hipEventCreateWithFlags(&h->ev_fork, sync_event_flags());
  if (e != hipSuccess) return (int)e;
  h->ev_chunk.resize(h->d.n_layers + 2);
  for (auto& ev : h->ev_chunk) {
    e = hipEventCreateWithFlags(&ev, sync_event_flags());
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
int ensure_wside(SlamEngine* h) {
  if (h->wside && h->wside_cus_applied == h->wside_cus) return 0;
  hipError_t e;
  if (h->wside) {  // the mask changed: replace the stream
    (void)hipStreamSynchronize(h->wside);
    (void)hipStreamDestroy(h->wside);
    h->wside = nullptr;
  }
  if (h->wside_cus > 0) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int n = h->wside_cus > 256 ? 256 : h->wside_cus;
    for (int i = 0; i < n; ++i) mask[i >> 5] |= 1u << (i & 31);
    e = hipExtStreamCreateWithCUMask(&h->wside, 8, mask);
  } else {
    e = hipStreamCreateWithFlags(&h->wside, hipStreamNonBlocking);
  }
  if (e != hipSuccess) return (int)e;
  h->wside_cus_applied = h->wside_cus;
  if (h->ev_w.empty()) {
    h->ev_w.resize((size_t)(h->d.n_layers + 1) * 8);
    for (auto& ev : h->ev_w) {
      e = hipEventCreateWithFlags(&ev, sync_event_flags());
      if (e != hipSuccess) return (int)e;
    }
  }
  return 0;
}
// make `st` wait for chunk i of a pending overlapped optimizer step
int wait_chunk(SlamEngine* h, int i, hipStream_t st) {
  if (!h->opt_pending) return 0;
  return (int)hipStreamWaitEvent(st, h->ev_chunk[i], 0);
}
// ... for all of it (the chunks are recorded in order on one stream: the last event covers them)
int join_optimizer(SlamEngine* h, hipStream_t st) {
  if (!h->opt_pending) return 0;
  h->opt_pending = false;
  return (int)hipStreamWaitEvent(st, h->ev_chunk.back(), 0);
}

// make `st` wait for the pending writers of parameters in [lo, hi); the stall is bracketed by two timing events
int wait_params(SlamEngine* h, int64_t lo, int64_t hi, hipStream_t st) {
  for (size_t i = 0; i < h->pwaits.size();) {
    SlamEngine::ParamWait& w = h->pwaits[i];
    if (w.lo < hi && lo < w.hi) {
      hipError_t r;
      // two timing events per wait are two more packets on the caller's stream: measurement runs only. The pool is read and
      // reset by slam_param_wait_ms. A caller that reads rarely (logging_steps >> 80) fills it: the pairs that have completed
      // - all but the last step's - are folded into a running sum and their slots reused, so the reported total stays complete;
      // a wait that still finds no slot is counted in pw_untimed (slam_param_wait_untimed) instead of vanishing
      if (h->time_param_waits && h->pw_used + 2 > 4096) {
        size_t keep = 0;
        for (size_t k = 0; k + 1 < h->pw_used; k += 2) {
          float ms = 0.f;
          if (hipEventQuery(h->pw_ev[k + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->pw_ev[k], h->pw_ev[k + 1]) == hipSuccess) {
            h->pw_acc_ms += ms;
          } else {  // still in flight: keep the pair (swap it to the front)
            std::swap(h->pw_ev[keep], h->pw_ev[k]);
            std::swap(h->pw_ev[keep + 1], h->pw_ev[k + 1]);
            keep += 2;
          }
        }
        h->pw_used = keep;
        if (h->pw_used + 2 > 4096) ++h->pw_untimed;
      }
      if (h->time_param_waits && h->pw_used + 2 <= 4096) {
        if (h->pw_used + 2 > h->pw_ev.size()) {
          for (int k = 0; k < 2; ++k) {
            hipEvent_t e;
            r = hipEventCreate(&e);
            if (r != hipSuccess) return (int)r;
            h->pw_ev.push_back(e);
          }
        }
        r = hipEventRecord(h->pw_ev[h->pw_used], st);
        if (r == hipSuccess) r = hipStreamWaitEvent(st, w.ev, 0);
        if (r == hipSuccess) r = hipEventRecord(h->pw_ev[h->pw_used + 1], st);
        h->pw_used += 2;
      } else {
        r = hipStreamWaitEvent(st, w.ev, 0);
      }
      if (r != hipSuccess) return (int)r;
      h->pwaits.erase(h->pwaits.begin() + i);
    } else {
      ++i;
    }
  }
  return 0;
}
int join_params(SlamEngine* h, hipStream_t st) { return h->pwaits.empty() ? 0 : wait_params(h, 0, h->n_params, st); }

// AdamW over the parameters [chunk c of the model]: c = 0 embedding, 1 + l = decoder layer l, L + 1 = final norm; c < 0 = all.
// With transposed weight images bound, every matrix goes through the tile kernel that writes its transposed image in the
// same pass (no separate transpose launch), the vectors between them through the strided kernel.
// mode 0: fp32 master + fp32 moments, 1: fp32 master + bf16 moments, 2: bf16 parameters + bf16 moments.
int adamw_model(SlamEngine* h, int mode, float* master, void* m, void* v, const float* norm_out, double lr, double b1, double b2,
                double eps, double wd, int step, int zero_grad, int chunk, hipStream_t st) {
  const SlamModelDesc& d = h->d;
  const int L = d.n_layers, H = d.hidden, I = d.intermediate, HD = d.n_heads * d.head_dim;
  bf16_t* P = h->params;
  bf16_t* Pt = h->params_t;
  const int g16 = h->gfinal == 2;  // the last backward kept its final values in bf16 only
  char* G = g16 ? (char*)h->g16 : (char*)h->grads;
  const size_t gsz = g16 ? 2 : 4;
  const size_t esz = mode == 0 ? 4 : 2;
  auto mat = [&](int64_t off, int R, int C, int batch) -> int {
    return adamw_tiles(mode, master ? master + off : nullptr, P + off, Pt + off, G + off * gsz, g16, (char*)m + off * esz,
                       (char*)v + off * esz, R, C, batch, (size_t)h->layer_stride, norm_out, lr, b1, b2, eps, wd, step, zero_grad, st);
  };
  auto vec = [&](int64_t off, size_t n, int batch) -> int {
    return adamw_strided(mode, master ? master + off : nullptr, P + off, G + off * gsz, g16, (char*)m + off * esz, (char*)v + off * esz, n,
                         batch, (size_t)h->layer_stride, norm_out, lr, b1, b2, eps, wd, step, zero_grad, st);
  };
  int r = 0;
  if (chunk < 0 || chunk == 0) r = mat(h->off_embed, h->vpad, H, 1);
  if (r) return r;
  const int l0 = chunk < 0 ? 0 : chunk - 1, nl = chunk < 0 ? L : 1;
  if (chunk < 0 || (chunk >= 1 && chunk <= L)) {
    const LayerOff& o = h->lo[l0];
    if ((r = vec(o.ln1, (size_t)H, nl))) return r;
    if ((r = mat(o.wqkv, h->QKV, H, nl))) return r;
    if ((r = vec(o.bqkv, (size_t)h->QKV, nl))) return r;
    if ((r = mat(o.wo, H, HD, nl))) return r;
    if ((r = vec(o.ln2, (size_t)H, nl))) return r;
    if ((r = mat(o.wgu, 2 * I, H, nl))) return r;
    if ((r = mat(o.wd, H, I, nl))) return r;
  }
  if (chunk < 0 || chunk == L + 1) r = vec(h->off_norm, (size_t)H, 1);
  return r;
}

}  // namespace

extern "C" {

const char* slam_version(void) { return "slam-engine gfx950 r6"; }

int slam_engine_create(const SlamModelDesc* desc, SlamEngine** out) {
  if (!desc || !out) return SLAM_EINVAL;
  const SlamModelDesc& d = *desc;
  if ((d.head_dim != 64 && d.head_dim != 128) || d.n_heads <= 0 || d.n_kv_heads <= 0 || d.n_heads % d.n_kv_heads) return SLAM_EINVAL;
  if (d.hidden % 8 || d.hidden > 2048 || d.intermediate % GU_BLK || d.vocab <= 0) return SLAM_EINVAL;
  if (d.n_layers <= 0) return SLAM_EINVAL;
  SlamEngine* e = new SlamEngine();
  e->d = d;
  e->QKV = (d.n_heads + 2 * d.n_kv_heads) * d.head_dim;
  e->vpad = d.vocab <= VPAD_SMALL ? VPAD_SMALL : ((d.vocab + 255) / 256) * 256;  // 256: the LM-head GEMM can take the 256 x 256 kernel
  e->fuse_swiglu = (d.hidden % 64 == 0) && ((2 * d.intermediate) % 128 == 0);
  int64_t off = 0;
  e->off_embed = off;
  add_tensor(e, "embed", off, e->vpad, d.hidden);
  e->lo.resize(d.n_layers);
  for (int l = 0; l < d.n_layers; ++l) {
    std::string p = "layers." + std::to_string(l) + ".";
    LayerOff& o = e->lo[l];
    o.ln1 = off;  add_tensor(e, p + "ln1", off, d.hidden, 1);
    o.wqkv = off; add_tensor(e, p + "wqkv", off, e->QKV, d.hidden);
    o.bqkv = off; add_tensor(e, p + "bqkv", off, e->QKV, 1);
    o.wo = off;   add_tensor(e, p + "wo", off, d.hidden, d.n_heads * d.head_dim);
    o.ln2 = off;  add_tensor(e, p + "ln2", off, d.hidden, 1);
    o.wgu = off;  add_tensor(e, p + "wgu", off, 2 * d.intermediate, d.hidden);
    o.wd = off;   add_tensor(e, p + "wd", off, d.hidden, d.intermediate);
  }
  e->layer_stride = d.n_layers > 1 ? e->lo[1].ln1 - e->lo[0].ln1 : (off - e->lo[0].ln1);
  e->off_norm = off;
  add_tensor(e, "norm", off, d.hidden, 1);
  e->n_params = off;
  *out = e;
  return SLAM_OK;
}

void slam_engine_destroy(SlamEngine* h) {
  if (h) (void)slam_comm_destroy(h);  // a communicator the caller did not release (needs RCCL's symbols: not the destructor's job)
  delete h;
}
const char* slam_last_error(SlamEngine* h) { return h ? h->err.c_str() : "null engine"; }
int64_t slam_param_count(SlamEngine* h) { return h ? h->n_params : 0; }
int32_t slam_tensor_count(SlamEngine* h) { return h ? (int32_t)h->tensors.size() : 0; }
int slam_tensor_info(SlamEngine* h, int32_t i, SlamTensorInfo* out) {
  if (!h || !out || i < 0 || i >= (int32_t)h->tensors.size()) return SLAM_EINVAL;
  *out = h->tensors[i];
  return SLAM_OK;
}
int slam_bind_params(SlamEngine* h, void* params_bf16, float* grads_f32) {
  if (!h || !params_bf16) return SLAM_EINVAL;
  h->params = (bf16_t*)params_bf16;
  h->grads = grads_f32;
  return SLAM_OK;
}
int slam_refresh_transposed(SlamEngine* h, slam_stream_t stream) {
  if (!h) return SLAM_EINVAL;
  if (!h->params_t) return SLAM_OK;
  if (!h->params) return h->fail(SLAM_ESTATE, "bind params first");
  hipStream_t st = (hipStream_t)stream;
  CK(join_optimizer(h, st));
  CK(join_params(h, st));
  h->params_t_dirty = false;
  const SlamModelDesc& d = h->d;
  const bf16_t* P = h->params;
  bf16_t* Pt = h->params_t;
  const int H = d.hidden, I = d.intermediate, HD = d.n_heads * d.head_dim;
  CK(transpose_bf16(P + h->off_embed, Pt + h->off_embed, h->vpad, H, 1, 0, st));
  // every layer has the same shapes at a constant stride: one launch per weight kind, grid.z = layers
  const LayerOff& o = h->lo[0];
  const int L = d.n_layers;
  const size_t ls = (size_t)h->layer_stride;
  CK(transpose_bf16(P + o.wqkv, Pt + o.wqkv, h->QKV, H, L, ls, st));
  CK(transpose_bf16(P + o.wo, Pt + o.wo, H, HD, L, ls, st));
  CK(transpose_bf16(P + o.wgu, Pt + o.wgu, 2 * I, H, L, ls, st));
  CK(transpose_bf16(P + o.wd, Pt + o.wd, H, I, L, ls, st));
  return SLAM_OK;
}
int slam_bind_params_t(SlamEngine* h, void* params_t_bf16) {
  if (!h) return SLAM_EINVAL;
  const SlamModelDesc& d = h->d;
  if (params_t_bf16 && ((d.hidden & 63) || (d.intermediate & 63) || (h->QKV & 63) || ((d.n_heads * d.head_dim) & 63)))
    return h->fail(SLAM_EINVAL, "transposed weight images need dims that are multiples of 64");
  h->params_t = (bf16_t*)params_t_bf16;
  return SLAM_OK;
}
size_t slam_workspace_bytes(SlamEngine* h, int64_t max_tokens) {
  if (!h || max_tokens <= 0) return 0;
  GemmTuneScope tune_scope(&h->gemm_tune);
  SlamEngine tmp;
  tmp.d = h->d;
  tmp.QKV = h->QKV;
  tmp.vpad = h->vpad;
  tmp.n_params = h->n_params;
  return carve(&tmp, nullptr, max_tokens);
}
int slam_bind_workspace(SlamEngine* h, void* ws, size_t bytes, int64_t max_tokens) {
  if (!h || !ws || max_tokens <= 0) return SLAM_EINVAL;
  if (((uintptr_t)ws) & 255) return h->fail(SLAM_EINVAL, "workspace must be 256-byte aligned");
  size_t need = slam_workspace_bytes(h, max_tokens);
  if (bytes < need) return h->fail(SLAM_ENOMEM, "workspace too small");
  carve(h, (char*)ws, max_tokens);
  h->ws = (char*)ws;
  h->ws_bytes = bytes;
  h->max_tokens = max_tokens;
  h->have_fwd = false;
  return SLAM_OK;
}
int slam_set_option(SlamEngine* h, const char* key, int64_t value) {
  if (!key) return SLAM_EINVAL;
  if (!strncmp(key, "gemm_", 5)) {  // kernel-selection knobs: this engine's (h) or the process default's (h = NULL)
    const int r = gemm_tune_set(h ? &h->gemm_tune : gemm_default_tune(), key, (long)value);
    if (r > 0) return SLAM_OK;
    return h ? h->fail(SLAM_EINVAL, std::string(r < 0 ? "value out of range for option " : "unknown option ") + key) : SLAM_EINVAL;
  }
  if (!strcmp(key, "attn_jq") || !strcmp(key, "attn_kw") || !strcmp(key, "attn_nch") || !strcmp(key, "attn_prio")) {
    // with an engine: that engine's launches (takes effect at its next forward, which rebuilds the attention plan);
    // without: the process default picked up by the single-op entry points and by engines created afterwards
    AttnTune t = h ? h->attn_tune : attn_default_tune();
    (key[5] == 'j' ? t.jq : key[5] == 'k' ? t.kw : key[5] == 'n' ? t.nch : t.prio) = (int)value;
    if (h) { h->attn_tune = t; h->have_fwd = false; } else attn_set_default_tune(t);
    return SLAM_OK;
  }
  if (!strcmp(key, "overlap_adamw") && h) { h->overlap_adamw = value != 0; return SLAM_OK; }
  if (!strcmp(key, "bwd_wgrad_stream") && h) { h->wgrad_stream = value != 0; return SLAM_OK; }
  if (!strcmp(key, "bwd_aux_side") && h) { h->aux_side = value != 0; return SLAM_OK; }
  if (!strcmp(key, "bwd_wgrad_cus") && h) { h->wside_cus = (int)value; return SLAM_OK; }
  if (!strcmp(key, "time_param_waits") && h) { h->time_param_waits = value != 0; return SLAM_OK; }
  if (!strcmp(key, "time_gateup") && h) {
    if (value && h->tg_ev.empty()) {
      h->tg_ev.resize((size_t)2 * h->d.n_layers);
      for (auto& e : h->tg_ev)
        if (hipEventCreate(&e) != hipSuccess) { h->tg_ev.clear(); return h->fail(SLAM_ESTATE, "hipEventCreate failed"); }
    }
    h->time_gateup = value != 0;
    return SLAM_OK;
  }
  if (!strcmp(key, "time_families") && h) { h->time_families = value != 0; if (!value) h->fam_marks.clear(); return SLAM_OK; }
  if (!strcmp(key, "grad_overwrite_next") && h) { h->overwrite_next = value != 0; return SLAM_OK; }
  if (!strcmp(key, "grad_norm_partials") && h) { h->norm_partials = value != 0; return SLAM_OK; }
  if (!strcmp(key, "grad_final_next") && h) {
    if (value < 0 || value > 2) return h->fail(SLAM_EINVAL, "grad_final_next takes 0, 1 or 2");
    h->final_next = (int)value;
    return SLAM_OK;
  }
  if (!strcmp(key, "fuse_swiglu") && h) { h->fuse_swiglu = value != 0; return SLAM_OK; }
  if (!strcmp(key, "fuse_dswiglu") && h) { h->fuse_dswiglu = value != 0; return SLAM_OK; }
  if (!strcmp(key, "fuse_adamw_t") && h) { h->fuse_adamw_t = value != 0; return SLAM_OK; }
  return h ? h->fail(SLAM_EINVAL, std::string("unknown option ") + key) : SLAM_EINVAL;
}

int slam_forward(SlamEngine* h, const int64_t* ids, const int64_t* labels, const int64_t* position_ids,
                 const int32_t* seg_start, const int32_t* seg_end, int32_t B, int32_t T, double num_items,
                 float* loss_out, void* logits_out, slam_stream_t stream) {
  if (!h || !ids || B <= 0 || T <= 0) return SLAM_EINVAL;
  if (!h->params || !h->ws) return h->fail(SLAM_ESTATE, "bind params and workspace first");
  const int64_t M64 = (int64_t)B * T;
  if (M64 > h->max_tokens) return h->fail(SLAM_ENOMEM, "B*T exceeds bound workspace tokens");
  if ((seg_start == nullptr) != (seg_end == nullptr)) return h->fail(SLAM_EINVAL, "seg_start/seg_end both or none");
  if (labels && !loss_out) return h->fail(SLAM_EINVAL, "labels given without loss_out");
  const int M = (int)M64;
  const SlamModelDesc& d = h->d;
  hipStream_t st = (hipStream_t)stream;
  const int H = d.hidden, I = d.intermediate, L = d.n_layers, nH = d.n_heads, nKV = d.n_kv_heads;
  const bf16_t* P = h->params;
  h->have_fwd = false;
  GemmTuneScope tune_scope(&h->gemm_tune);

  if (seg_start) {
    h->cur_seg_s = seg_start;
    h->cur_seg_e = seg_end;
  } else {
    seg_fill_kernel<<<(M + 255) / 256, 256, 0, st>>>(h->seg_s, h->seg_e, M, T);
    h->cur_seg_s = h->seg_s;
    h->cur_seg_e = h->seg_e;
  }
  if (h->time_families) h->fam_marks.clear();  // the marks of a step = its last forward + backward
  CK(attn_plan(h->cur_seg_s, h->cur_seg_e, M, d.head_dim, h->attn_tune, h->attn_plan_buf, st));
  // queries are stored pre-scaled by head_dim^-0.5 * log2(e) (folded into their rotation tables: one rounding), so the
  // attention kernels' scores leave the matrix pipe in the exp2 domain
  const float qscale = 1.44269504088896340736f / sqrtf((float)d.head_dim);
  CK(rope_table(position_ids, M, T, d.head_dim, d.rope_theta, h->cosb, h->sinb, h->cosq, h->sinq, qscale, st));
  CK(wait_chunk(h, 0, st));
  CK(wait_params(h, h->off_embed, h->lo[0].ln1, st));
  CK(embed_fwd(ids, P + h->off_embed, h->hs[0], M, H, d.vocab, st));
  for (int l = 0; l < L; ++l) {
    const LayerOff& o = h->lo[l];
    LayerAct& a = h->la[l];
    CK(wait_chunk(h, 1 + l, st));
    CK(wait_params(h, o.ln1, o.ln1 + h->layer_stride, st));
    TK(F_NORM_FWD, st, rmsnorm_fwd(h->hs[l], P + o.ln1, a.x1, a.rstd1, M, H, d.rms_eps, st));
    if (d.head_dim == 64 && (H % 64 == 0) && (h->QKV % 128 == 0)) {  // bias + RoPE fused into the projection epilogue
      TK(F_QKV_FWD, st, gemm_nt_rope(a.x1, P + o.wqkv, a.qkv, P + o.bqkv, h->cosb, h->sinb, h->cosq, h->sinq, nH, nH + nKV, M, h->QKV, H, st));
    } else {
      const int slot = fam_begin(h, F_QKV_FWD, st);
      CK(gemm_nt(a.x1, P + o.wqkv, a.qkv, P + o.bqkv, nullptr, M, h->QKV, H, st));
      CK(rope_apply(a.qkv, h->QKV, M, nH + nKV, d.head_dim, h->cosb, h->sinb, 0, st, nH, qscale));
      fam_end(h, slot, st);
    }
    TK(F_ATTN_FWD, st, attn_fwd(a.qkv, a.o, a.lse, h->cur_seg_s, h->attn_plan_buf, h->attn_tune, M, nH, nKV, d.head_dim, st));
    TK(F_O_FWD, st, gemm_nt(a.o, P + o.wo, a.hmid, nullptr, h->hs[l], M, H, nH * d.head_dim, st));
    TK(F_NORM_FWD, st, rmsnorm_fwd(a.hmid, P + o.ln2, a.x2, a.rstd2, M, H, d.rms_eps, st));
    const bool timed = h->time_gateup && h->tg_ev.size() == (size_t)(2 * L);
    if (timed) CK((int)hipEventRecord(h->tg_ev[2 * l], st));
    {
      const int slot = fam_begin(h, F_GATEUP_FWD, st);
      if (h->fuse_swiglu) {
        CK(gemm_nt_swiglu(a.x2, P + o.wgu, a.gu, a.act, M, 2 * I, H, st));
      } else {
        CK(gemm_nt(a.x2, P + o.wgu, a.gu, nullptr, nullptr, M, 2 * I, H, st));
        CK(swiglu_fwd(a.gu, a.act, M, I, GU_BLK, st));
      }
      fam_end(h, slot, st);
    }
    if (timed) CK((int)hipEventRecord(h->tg_ev[2 * l + 1], st));
    TK(F_DOWN_FWD, st, gemm_nt(a.act, P + o.wd, h->hs[l + 1], nullptr, a.hmid, M, H, I, st));
  }
  CK(join_optimizer(h, st));
  CK(join_params(h, st));
  TK(F_NORM_FWD, st, rmsnorm_fwd(h->hs[L], P + h->off_norm, h->hf, h->rstdf, M, H, d.rms_eps, st));
  const int VP = h->vpad;
  TK(F_HEAD_FWD, st, gemm_nt(h->hf, P + h->off_embed, h->logits, nullptr, nullptr, M, VP, H, st));
  h->have_loss = false;
  if (logits_out) CK(copy_cols(h->logits, VP, (bf16_t*)logits_out, d.vocab, M, d.vocab, st));
  if (labels) {
    TK(F_LOSS, st, cross_entropy(h->logits, labels, num_items, h->dlogits, h->row_loss, h->scal + 0, h->scal + 1, B, T, VP,
                     d.vocab, h->logit_mask, st));
    CK((int)hipMemcpyAsync(loss_out, h->scal + 1, sizeof(float), hipMemcpyDeviceToDevice, st));
    h->have_loss = true;
  }
  h->B = B;
  h->T = T;
  h->last_ids = ids;
  h->have_fwd = true;
  return SLAM_OK;
}

int slam_backward(SlamEngine* h, float grad_scale, int32_t bucket_layers, slam_bucket_cb cb, void* user,
                  slam_stream_t stream) {
  if (!h) return SLAM_EINVAL;
  if (!h->have_fwd || !h->have_loss) return h->fail(SLAM_ESTATE, "backward needs a forward with labels");
  if (!h->grads) return h->fail(SLAM_ESTATE, "no gradient buffer bound");
  const SlamModelDesc& d = h->d;
  hipStream_t st = (hipStream_t)stream;
  CK(join_optimizer(h, st));
  CK(join_params(h, st));
  h->ag_ev_used = 0;  // every parameter arrival has been waited for: the pool is free for the next step's gathers
  if (h->params_t_dirty) CK(slam_refresh_transposed(h, stream));
  const int M = h->B * h->T;
  const int H = d.hidden, I = d.intermediate, L = d.n_layers, nH = d.n_heads, nKV = d.n_kv_heads;
  const int HD = nH * d.head_dim;
  const bf16_t* P = h->params;
  float* G = h->grads;

  const bf16_t* Pt = h->params_t;
  // dX[M,K] = dY[M,N] W[N,K]: with a transposed image W^T[K,N] it is the contraction-contiguous form
  auto dgrad = [&](const bf16_t* dY, int64_t woff, bf16_t* dX, int N, int K) -> int {
    return Pt ? gemm_nt(dY, Pt + woff, dX, nullptr, nullptr, M, K, N, st) : gemm_nn(dY, P + woff, dX, nullptr, M, N, K, st);
  };
  const int VP = h->vpad;
  if (grad_scale != 1.0f) CK(scale_bf16(h->dlogits, (size_t)M * VP, grad_scale, st));
  // tied head: dE += dlogits^T hf ; dhf = dlogits E
  // first micro-batch of an optimizer step: every gradient tensor is written exactly once below (the embedding
  // twice: head first, gather side second), so it may be STORED instead of accumulated and the buffer needs no
  // zeroing pass (4 B/param written by AdamW + 4 B/param re-read by the wgrad epilogues)
  const int acc = h->overwrite_next ? 0 : 1;
  h->overwrite_next = false;
  // bf16 communication image of the gradients (slam_set_grad_image): written by the SAME kernels that store the final fp32
  // values (unsplit weight-gradient tiles, slab reduces, the norm / bias finish kernel) - no conversion pass over the buffer
  bf16_t* const IMG = h->grad_img;
  h->grad_img = nullptr;
  h->last_grad_img = IMG;
  auto img = [&](int64_t off) -> bf16_t* { return IMG ? IMG + off : nullptr; };
  // last backward of an optimizer step ("grad_final_next"): sum-of-squares partials from the final-value stores, and
  // (mode 2) final values in the bf16 image only. Every launch that stores final values takes its slots in launch order.
  const int fin = h->final_next;
  h->final_next = 0;
  h->gfinal = 0;
  h->g16 = nullptr;
  if (fin == 2 && !IMG) return h->fail(SLAM_ESTATE, "grad_final_next = 2 needs slam_set_grad_image before the backward");
  // known from here on: the bucket callbacks below run INSIDE this call, and the engine-side exchange they may start
  // (slam_allreduce_grads_async / slam_reduce_scatter_grads_async) asks where the gradients live
  h->gfinal = fin;
  h->g16 = fin == 2 ? IMG : nullptr;
  const bool partials = fin && !cb && h->norm_partials;  // with a bucket callback the gradients are about to be exchanged: partials of the local ones are of no use
  h->gn_valid = false;
  if (partials) {
    CK((int)hipMemsetAsync(h->gn_part, 0, h->gn_cap * sizeof(float), st));  // blocks without a final store leave their slot alone
    h->gn_used = 0;
  }
  GradSink sink_store;
  auto sink = [&]() -> GradSink* {  // the slots from gn_used on; take(r) after the launch
    if (!fin) return nullptr;
    sink_store.img_only = fin == 2;
    sink_store.sumsq = partials ? h->gn_part + h->gn_used : nullptr;
    sink_store.cap = (int)(h->gn_cap - h->gn_used);
    sink_store.used = 0;
    return &sink_store;
  };
  auto take = [&](int r) -> int {
    if (partials && r == 0) h->gn_used += (size_t)sink_store.used;
    return r;
  };

  // weight-gradient launches: on the main stream, or (bwd_wgrad_stream) on the side stream `ws` behind an event that the
  // main stream records once their operands exist. Every cross-stream edge costs the recording AND the waiting stream a
  // barrier packet (~6 us of queue bubble each, measured in the kernel trace), so the schedule keeps them few:
  //  * no side -> main edges at all: every gradient a weight-gradient GEMM reads lives in a buffer nothing overwrites
  //    until the next forward - the gradient of hs[l] goes into layer l's (dead) hmid buffer, the gradient of hmid[l]
  //    into the (dead) hs[l+1] buffer, d(qkv) of layer l into layer l+1's (dead) qkv buffer; all three are buffers only
  //    main-stream kernels read, and only earlier in this backward;
  //  * main -> side: one event per weight-gradient GEMM, recorded as soon as its operands exist. Handing them over in
  //    batches (two or one event per layer) was measured and is worse (+0.8 / +1.5 ms per Slam-358M step): the side stream
  //    needs its work as early as it can have it.
  const bool two = h->wgrad_stream != 0;
  if (two) CK(ensure_wside(h));
  GemmTuneScope tune_scope(&h->gemm_tune);
  struct SharedGuard {  // dgrad launches of this call may plan for a GPU they share with the wgrad stream
    GemmTune* t;
    SharedGuard(GemmTune* t_, int on) : t(t_) { t->shared = on; }
    ~SharedGuard() { t->shared = 0; }
  } shared_guard(&h->gemm_tune, two ? 1 : 0);
  hipStream_t ws = two ? h->wside : st;
  int ev_used = 0;
  auto edge = [&](hipStream_t from, hipStream_t to) -> int {  // `to` continues after everything enqueued on `from` so far
    if (from == to) return 0;
    if ((size_t)ev_used >= h->ev_w.size()) return SLAM_ESTATE;
    hipEvent_t e = h->ev_w[ev_used++];
    hipError_t r = hipEventRecord(e, from);
    if (r != hipSuccess) return (int)r;
    return (int)hipStreamWaitEvent(to, e, 0);
  };
  auto fork = [&]() -> int { return two ? edge(st, ws) : 0; };
  const bool aux = two && h->aux_side != 0;
  // dW (+)= a^T b on a weight-gradient stream
  // is_final: the launch stores the tensor's final values (everything but the head's half of the tied embedding gradient)
  auto wgrad = [&](int fam, const bf16_t* a, const bf16_t* b, float* g, int n, int k, bf16_t* gi, bool is_final = true,
                   bool handed_over = false) -> int {  // handed_over: ws already waits for everything enqueued on st so far
    if (!handed_over)
      if (int r = fork()) return r;
    const int slot = fam_begin(h, fam, ws);
    const int r = take(gemm_tn(a, b, g, acc, M, n, k, n, k, h->gemm_ws, h->gemm_ws_bytes, ws, two ? 1 : 0, gi, is_final ? sink() : nullptr));
    fam_end(h, slot, ws);
    return r;
  };

  CK(wgrad(F_HEAD_WGRAD, h->dlogits, h->hf, G + h->off_embed, VP, H, nullptr, false));  // not final: the gather side adds to it below
  TK(F_HEAD_DGRAD, st, dgrad(h->dlogits, h->off_embed, h->dx, VP, H));
  bf16_t* dh = h->dh_a;  // grad wrt hs[l+1]
  TK(F_NORM_BWD, st, take(rmsnorm_bwd(h->dx, h->hs[L], P + h->off_norm, h->rstdf, nullptr, dh, G + h->off_norm, acc, h->part_ws, M, H, st, img(h->off_norm), sink())));

  const int bl = bucket_layers > 0 ? bucket_layers : L;
  int64_t bucket_end = h->n_params;  // exclusive end of the not-yet-reported range
  int fin_hi = L;                    // layers >= fin_hi have their norm/bias partial slabs finished
  for (int l = L - 1; l >= 0; --l) {
    const LayerOff& o = h->lo[l];
    LayerAct& a = h->la[l];
    bf16_t* dh2 = h->hs[l + 1];                              // grad wrt hmid[l]: hs[l+1] was last read by the norm backward above it
    bf16_t* dqkv = l + 1 < L ? h->la[l + 1].qkv : h->dqkv;   // layer l+1's q|k|v were last read by its attention backward
    // MLP
    CK(wgrad(F_WD_WGRAD, dh, a.act, G + o.wd, H, I, img(o.wd)));
    {
      const int slot = fam_begin(h, F_DOWN_DGRAD, st);
      if (Pt && h->fuse_dswiglu && (I % 128 == 0) && (H % 64 == 0)) {
        CK(gemm_nt_dswiglu(dh, Pt + o.wd, a.gu, M, I, H, st));  // d(act) stays in registers; a.gu -> d(gate|up)
      } else {
        CK(dgrad(dh, o.wd, h->dact, H, I));
        CK(swiglu_bwd(a.gu, h->dact, M, I, GU_BLK, st));  // a.gu now holds d(gate|up)
      }
      fam_end(h, slot, st);
    }
    CK(wgrad(F_WGU_WGRAD, a.gu, a.x2, G + o.wgu, 2 * I, H, img(o.wgu)));
    TK(F_GATEUP_DGRAD, st, dgrad(a.gu, o.wgu, h->dx, 2 * I, H));
    TK(F_NORM_BWD, st, rmsnorm_bwd(h->dx, a.hmid, P + o.ln2, a.rstd2, dh, dh2, nullptr, 1, h->ln_part + (size_t)(2 * l + 1) * h->ln_ps, M, H, st));
    // attention
    CK(wgrad(F_WO_WGRAD, dh2, a.o, G + o.wo, H, HD, img(o.wo)));
    TK(F_O_DGRAD, st, dgrad(dh2, o.wo, h->d_o, H, HD));
    TK(F_ATTN_BWD, st, attn_bwd(a.qkv, a.o, h->d_o, a.lse, h->dsum, h->nlse, dqkv, h->dkv_part, h->cur_seg_s, h->cur_seg_e, h->attn_plan_buf, h->attn_tune, h->cosb, h->sinb,
                M, nH, nKV, d.head_dim, st));  // dq / dk come out already rotated back
    if (aux) {  // behind the same hand-over as the Wqkv gradient: both read d(qkv)
      CK(fork());
      CK(colsum_bf16(dqkv, h->QKV, M, h->QKV, nullptr, 1, h->bias_part + (size_t)l * h->bias_ps, ws));
    } else {
      CK(colsum_bf16(dqkv, h->QKV, M, h->QKV, nullptr, 1, h->bias_part + (size_t)l * h->bias_ps, st));
    }
    CK(wgrad(F_WQKV_WGRAD, dqkv, a.x1, G + o.wqkv, h->QKV, H, img(o.wqkv), true, aux));
    TK(F_QKV_DGRAD, st, dgrad(dqkv, o.wqkv, h->dx, h->QKV, H));
    dh = a.hmid;  // grad wrt hs[l]: hmid[l] was last read by the ln2 backward above
    TK(F_NORM_BWD, st, rmsnorm_bwd(h->dx, h->hs[l], P + o.ln1, a.rstd1, dh2, dh, nullptr, 1, h->ln_part + (size_t)(2 * l) * h->ln_ps, M, H, st));
    // bucket boundaries: every `bl` layers from the top, and after each of the last two layers so that the
    // final all-reduce (exposed behind the end of backward) only carries layer 0 + the embedding
    const bool boundary = cb && l > 0 && ((((L - l) % bl) == 0) || l <= 2);
    if (l == 0 || boundary) {
      // the layers [l, fin_hi) are complete: finish their norm / bias partial slabs in three launches
      const int cnt = fin_hi - l;
      const int nbl = rmsnorm_bwd_blocks(M), nbc = colsum_blocks(M);
      hipStream_t fs = aux ? ws : st;  // aux: after the norm backward above (the hand-over) and the column sums already on ws
      if (aux) CK(fork());
      CK(take(colsum_finish_many(h->ln_part + (size_t)(2 * l) * h->ln_ps, 2 * h->ln_ps, nbl, H, G + o.ln1, (size_t)h->layer_stride, cnt, acc, fs, img(o.ln1), sink())));
      CK(take(colsum_finish_many(h->ln_part + (size_t)(2 * l + 1) * h->ln_ps, 2 * h->ln_ps, nbl, H, G + o.ln2, (size_t)h->layer_stride, cnt, acc, fs, img(o.ln2), sink())));
      CK(take(colsum_finish_many(h->bias_part + (size_t)l * h->bias_ps, h->bias_ps, nbc, h->QKV, G + o.bqkv, (size_t)h->layer_stride, cnt, acc, fs, img(o.bqkv), sink())));
      fin_hi = l;
    }
    if (boundary) {
      // the side stream is in order: its last launch of layer l covers every wgrad of the range. Order it after the
      // finish kernels above as well and hand IT to the consumer (slam_bucket_stream): main does not stall here.
      if (!aux) CK(fork());  // aux: the finish kernels ARE on the side stream, behind a hand-over of their own
      h->bucket_stream = two ? ws : nullptr;
      cb(user, o.ln1, bucket_end - o.ln1);
      h->bucket_stream = nullptr;
      bucket_end = o.ln1;
    }
  }
  // gather-side embedding gradient (padding_idx row suppressed): small vocabularies run it as
  // dE += onehot(ids)^T dh0 on the wgrad GEMM, large ones as a token-ordered scatter; both deterministic
  // (on the wgrad stream: ordered after the head's contribution to the same rows)
  CK(fork());
  {
    const int slot = fam_begin(h, F_EMBED_WGRAD, ws);
    if (VP == VPAD_SMALL) {
      CK(onehot(h->last_ids, h->onehot, M, VP, d.vocab, d.pad_token_id, ws));
      CK(take(gemm_tn(h->onehot, dh, G + h->off_embed, 1, M, VP, H, VP, H, h->gemm_ws, h->gemm_ws_bytes, ws, two ? 1 : 0, img(h->off_embed), sink())));
    } else {
      CK(embed_bwd(h->last_ids, dh, G + h->off_embed, M, H, VP, d.vocab, d.pad_token_id, h->embed_ws, ws));
      // the scatter only touches the rows that occur in the batch (the others keep the head's contribution): this one tensor
      // gets its image from a conversion pass (which also emits its partials when the values kept are the rounded ones)
      const size_t ne = (size_t)VP * H;
      if (fin == 2 && !partials) {
        CK(f32_to_bf16(G + h->off_embed, IMG + h->off_embed, ne, ws));
      } else if (fin == 2) {
        const int slots = f32_to_bf16_sumsq_slots(ne);
        if (h->gn_used + (size_t)slots > h->gn_cap) return h->fail(SLAM_ESTATE, "gradient-norm partial slots exhausted");
        CK(f32_to_bf16_sumsq(G + h->off_embed, IMG + h->off_embed, ne, h->gn_part + h->gn_used, ws));
        h->gn_used += (size_t)slots;
      } else {
        if (IMG) CK(f32_to_bf16(G + h->off_embed, IMG + h->off_embed, ne, ws));
        if (partials) {  // fp32 values kept: their chunk sums (the tensor starts the buffer: chunk-aligned, its end is `n` here)
          const size_t slots = (ne + grad_chunk_elems() - 1) / grad_chunk_elems();
          if (h->gn_used + slots > h->gn_cap) return h->fail(SLAM_ESTATE, "gradient-norm partial slots exhausted");
          CK(grad_sumsq_chunks(G + h->off_embed, 0, ne, 0, ne, h->gn_part + h->gn_used, ws));
          h->gn_used += slots;
        }
      }
    }
    fam_end(h, slot, ws);
  }
  if (two) CK(edge(ws, st));  // join: everything after slam_backward on `stream` sees complete gradients
  if (cb) cb(user, 0, bucket_end);
  h->have_loss = false;  // a.gu was consumed; a second backward needs a new forward
  h->gn_valid = partials;
  return SLAM_OK;
}

slam_stream_t slam_bucket_stream(SlamEngine* h) { return h ? (slam_stream_t)h->bucket_stream : nullptr; }

int slam_set_logit_mask(SlamEngine* h, const uint8_t* mask) {
  if (!h) return SLAM_EINVAL;
  h->logit_mask = mask;
  return SLAM_OK;
}
int32_t slam_padded_vocab(SlamEngine* h) { return h ? h->vpad : 0; }

int slam_seq_loglik(SlamEngine* h, const int64_t* labels, int32_t B, int32_t T, float* ll_out, float* cnt_out,
                    slam_stream_t stream) {
  if (!h || !labels || !ll_out || !cnt_out) return SLAM_EINVAL;
  if (!h->have_fwd || B != h->B || T != h->T) return h->fail(SLAM_ESTATE, "seq_loglik needs the matching forward");
  CK(seq_loglik(h->row_loss, labels, B, T, ll_out, cnt_out, (hipStream_t)stream));
  return SLAM_OK;
}

int slam_scale_loss_rows(SlamEngine* h, const float* seq_coef, int32_t B, int32_t T, slam_stream_t stream) {
  if (!h || !seq_coef) return SLAM_EINVAL;
  if (!h->have_loss || B != h->B || T != h->T) return h->fail(SLAM_ESTATE, "scale_loss_rows needs the matching forward with labels");
  CK(scale_rows_bf16(h->dlogits, seq_coef, B * T, T, h->vpad, (hipStream_t)stream));
  return SLAM_OK;
}

int slam_grad_norm(SlamEngine* h, float max_norm, float* norm_out, slam_stream_t stream) {
  if (!h || !norm_out) return SLAM_EINVAL;
  if (!h->grads || !h->ws) return h->fail(SLAM_ESTATE, "bind params and workspace first");
  CK(join_optimizer(h, (hipStream_t)stream));
  if (h->gn_valid) {  // the last backward emitted the partial sums of squares with its final-value stores: add them, in slot order
    CK(grad_norm_from_chunks(h->gn_part, h->gn_used, max_norm, norm_out, (hipStream_t)stream));
    return SLAM_OK;
  }
  // chunk sums over the buffer the gradients are in (the bf16 image after a "grad_final_next" = 2 backward whose buckets were exchanged)
  const int g16 = h->gfinal == 2;
  const size_t n = (size_t)h->n_params;
  CK(grad_sumsq_chunks(g16 ? (const void*)h->g16 : (const void*)h->grads, g16, n, 0, n, h->part_ws, (hipStream_t)stream));
  CK(grad_norm_from_chunks(h->part_ws, (n + grad_chunk_elems() - 1) / grad_chunk_elems(), max_norm, norm_out, (hipStream_t)stream));
  return SLAM_OK;
}

static int adamw_any(SlamEngine* h, int mode, float* master, void* m, void* v, const float* norm_out, double lr, double b1, double b2,
                     double eps, double wd, int32_t step, int32_t zero_grad, slam_stream_t stream) {
  if (!h || !m || !v || step < 1 || (mode != 2 && !master)) return SLAM_EINVAL;
  if (!h->grads || !h->params) return h->fail(SLAM_ESTATE, "bind params first");
  if (h->n_params & 7) return h->fail(SLAM_EINVAL, "parameter count must be a multiple of 8");
  hipStream_t st = (hipStream_t)stream;
  CK(join_optimizer(h, st));
  CK(join_params(h, st));
  const bool fused = h->params_t != nullptr && h->fuse_adamw_t;
  const int g16 = h->gfinal == 2;  // the last backward kept its final values in bf16 only (slam_set_grad_image's buffer)
  char* const G = g16 ? (char*)h->g16 : (char*)h->grads;
  const size_t gsz = g16 ? 2 : 4;
  // zeroing belongs to the fp32 accumulation buffer: the kernels cannot do it while they read the bf16 image
  if (g16 && zero_grad) CK((int)hipMemsetAsync(h->grads, 0, (size_t)h->n_params * sizeof(float), st));
  if (!h->overlap_adamw || mode != 0) {
    if (fused) {
      CK(adamw_model(h, mode, master, m, v, norm_out, lr, b1, b2, eps, wd, step, zero_grad, -1, st));
      h->params_t_dirty = false;
      return SLAM_OK;
    }
    if (mode == 0) CK(adamw(master, h->params, G, g16, (float*)m, (float*)v, (size_t)h->n_params, norm_out, lr, b1, b2, eps, wd, step, zero_grad, st));
    else if (mode == 2) CK(adamw_bf16(h->params, G, g16, (bf16_t*)m, (bf16_t*)v, (size_t)h->n_params, norm_out, lr, b1, b2, eps, wd, step, zero_grad, st));
    else CK(adamw_strided(1, master, h->params, G, g16, m, v, (size_t)h->n_params, 1, 0, norm_out, lr, b1, b2, eps, wd, step, zero_grad, st));
    return slam_refresh_transposed(h, stream);
  }
  // "overlap_adamw" (fp32 state): per-layer chunks on the engine's side stream; the next forward waits per layer
  CK(ensure_side(h));
  CK((int)hipEventRecord(h->ev_fork, st));
  CK((int)hipStreamWaitEvent(h->side, h->ev_fork, 0));
  const SlamModelDesc& d = h->d;
  const int L = d.n_layers, H = d.hidden, I = d.intermediate, HD = d.n_heads * d.head_dim;
  bf16_t* Pt = h->params_t;
  for (int c = 0; c < L + 2; ++c) {
    if (fused) {
      CK(adamw_model(h, 0, master, m, v, norm_out, lr, b1, b2, eps, wd, step, zero_grad, c, h->side));
    } else {
      const int64_t lo = c == 0 ? 0 : c <= L ? h->lo[c - 1].ln1 : h->off_norm;
      const int64_t hi = c == 0 ? h->lo[0].ln1 : c < L ? h->lo[c].ln1 : c == L ? h->off_norm : h->n_params;
      CK(adamw(master + lo, h->params + lo, G + lo * gsz, g16, (float*)m + lo, (float*)v + lo, (size_t)(hi - lo), norm_out, lr, b1, b2, eps, wd, step,
               zero_grad, h->side));
      if (Pt) {
        const bf16_t* P = h->params;
        if (c == 0) {
          CK(transpose_bf16(P + h->off_embed, Pt + h->off_embed, h->vpad, H, 1, 0, h->side));
        } else if (c <= L) {
          const LayerOff& o = h->lo[c - 1];
          CK(transpose_bf16(P + o.wqkv, Pt + o.wqkv, h->QKV, H, 1, 0, h->side));
          CK(transpose_bf16(P + o.wo, Pt + o.wo, H, HD, 1, 0, h->side));
          CK(transpose_bf16(P + o.wgu, Pt + o.wgu, 2 * I, H, 1, 0, h->side));
          CK(transpose_bf16(P + o.wd, Pt + o.wd, H, I, 1, 0, h->side));
        }
      }
    }
    CK((int)hipEventRecord(h->ev_chunk[c], h->side));
  }
  h->opt_pending = true;
  return SLAM_OK;
}

int slam_adamw_step(SlamEngine* h, float* master, float* m, float* v, const float* norm_out, double lr, double b1,
                    double b2, double eps, double wd, int32_t step, int32_t zero_grad, slam_stream_t stream) {
  return adamw_any(h, 0, master, m, v, norm_out, lr, b1, b2, eps, wd, step, zero_grad, stream);
}

int slam_adamw_step_bf16_moments(SlamEngine* h, float* master, void* exp_avg_bf16, void* exp_avg_sq_bf16, const float* norm_out,
                                 double lr, double b1, double b2, double eps, double wd, int32_t step, int32_t zero_grad,
                                 slam_stream_t stream) {
  return adamw_any(h, 1, master, exp_avg_bf16, exp_avg_sq_bf16, norm_out, lr, b1, b2, eps, wd, step, zero_grad, stream);
}

int slam_adamw_step_bf16(SlamEngine* h, void* exp_avg_bf16, void* exp_avg_sq_bf16, const float* norm_out, double lr,
                         double b1, double b2, double eps, double wd, int32_t step, int32_t zero_grad, slam_stream_t stream) {
  return adamw_any(h, 2, nullptr, exp_avg_bf16, exp_avg_sq_bf16, norm_out, lr, b1, b2, eps, wd, step, zero_grad, stream);
}

// ---- sharded optimizer (data-parallel "rs_ag": reduce-scatter gradients, update the owned 1/N shard, all-gather bf16
// parameters) ------------------------------------------------------------------------------------------------------
int64_t slam_grad_chunk_elems(void) { return grad_chunk_elems(); }
// element `offset` of the gradients, in the buffer the last backward left them in
static void* range_grads(SlamEngine* h, int64_t offset) {
  return h->gfinal == 2 ? (void*)(h->g16 + offset) : (void*)(h->grads + offset);
}

int slam_grad_sumsq_chunks(SlamEngine* h, int64_t offset, int64_t count, float* chunk_sums, slam_stream_t stream) {
  if (!h || !chunk_sums || offset < 0 || count < 0) return SLAM_EINVAL;
  if (!h->grads) return h->fail(SLAM_ESTATE, "no gradient buffer bound");
  const int g16 = h->gfinal == 2;  // the gradients (a reduced shard of them) are in the bf16 image
  int r = grad_sumsq_chunks(g16 ? (const void*)h->g16 : (const void*)h->grads, g16, (size_t)h->n_params, (size_t)offset, (size_t)count, chunk_sums,
                            (hipStream_t)stream);
  if (r < 0) return h->fail(SLAM_EINVAL, "range must start on a chunk boundary and end on one (or at the end of the buffer)");
  CK(r);
  return SLAM_OK;
}

int slam_grad_norm_from_chunks(SlamEngine* h, const float* chunk_sums, float max_norm, float* norm_out, slam_stream_t stream) {
  if (!h || !chunk_sums || !norm_out) return SLAM_EINVAL;
  const size_t nc = ((size_t)h->n_params + grad_chunk_elems() - 1) / grad_chunk_elems();
  CK(grad_norm_from_chunks(chunk_sums, nc, max_norm, norm_out, (hipStream_t)stream));
  return SLAM_OK;
}

int slam_adamw_range(SlamEngine* h, int64_t offset, int64_t count, float* master, float* m, float* v, const float* norm_out,
                     double lr, double b1, double b2, double eps, double wd, int32_t step, int32_t zero_grad,
                     slam_stream_t stream) {
  if (!h || !master || !m || !v || step < 1 || offset < 0 || count < 0 || offset + count > h->n_params || (offset & 3) || (count & 3))
    return SLAM_EINVAL;
  if (!h->grads || !h->params) return h->fail(SLAM_ESTATE, "bind params first");
  hipStream_t st = (hipStream_t)stream;
  CK(join_optimizer(h, st));
  if (count)
    CK(adamw(master, h->params + offset, range_grads(h, offset), h->gfinal == 2, m, v, (size_t)count, norm_out, lr, b1, b2, eps, wd, step, zero_grad, st));
  h->params_t_dirty = h->params_t != nullptr;
  return SLAM_OK;
}

int slam_adamw_range_bf16_moments(SlamEngine* h, int64_t offset, int64_t count, float* master, void* m_bf16, void* v_bf16,
                                  const float* norm_out, double lr, double b1, double b2, double eps, double wd, int32_t step,
                                  int32_t zero_grad, slam_stream_t stream) {
  if (!h || !master || !m_bf16 || !v_bf16 || step < 1 || offset < 0 || count < 0 || offset + count > h->n_params || (offset & 3) || (count & 3))
    return SLAM_EINVAL;
  if (!h->grads || !h->params) return h->fail(SLAM_ESTATE, "bind params first");
  hipStream_t st = (hipStream_t)stream;
  CK(join_optimizer(h, st));
  if (count)
    CK(adamw_strided(1, master, h->params + offset, range_grads(h, offset), h->gfinal == 2, m_bf16, v_bf16, (size_t)count, 1, 0, norm_out, lr, b1, b2,
                     eps, wd, step, zero_grad, st));
  h->params_t_dirty = h->params_t != nullptr;
  return SLAM_OK;
}

int slam_adamw_range_bf16(SlamEngine* h, int64_t offset, int64_t count, void* m_bf16, void* v_bf16, const float* norm_out,
                          double lr, double b1, double b2, double eps, double wd, int32_t step, int32_t zero_grad,
                          slam_stream_t stream) {
  if (!h || !m_bf16 || !v_bf16 || step < 1 || offset < 0 || count < 0 || offset + count > h->n_params || (offset & 7) || (count & 7))
    return SLAM_EINVAL;
  if (!h->grads || !h->params) return h->fail(SLAM_ESTATE, "bind params first");
  hipStream_t st = (hipStream_t)stream;
  CK(join_optimizer(h, st));
  if (count)
    CK(adamw_bf16(h->params + offset, range_grads(h, offset), h->gfinal == 2, (bf16_t*)m_bf16, (bf16_t*)v_bf16, (size_t)count, norm_out, lr, b1, b2,
                  eps, wd, step, zero_grad, st));
  h->params_t_dirty = h->params_t != nullptr;
  return SLAM_OK;
}

int slam_add_param_wait(SlamEngine* h, int64_t offset, int64_t count, void* event) {
  if (!h || !event || offset < 0 || count <= 0 || offset + count > h->n_params) return SLAM_EINVAL;
  h->pwaits.push_back({offset, offset + count, (hipEvent_t)event});
  return SLAM_OK;
}

int slam_gateup_launch_ms(SlamEngine* h, float* ms_out, int32_t n) {
  if (!h || !ms_out || n < h->d.n_layers) return SLAM_EINVAL;
  if (!h->time_gateup || h->tg_ev.size() != (size_t)(2 * h->d.n_layers)) return h->fail(SLAM_ESTATE, "set the time_gateup option and run a forward first");
  for (int l = 0; l < h->d.n_layers; ++l) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(h->tg_ev[2 * l + 1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->tg_ev[2 * l], h->tg_ev[2 * l + 1]);
    if (e != hipSuccess) return h->fail((int)e, "gate|up timing events are not recorded");
    ms_out[l] = ms;
  }
  return SLAM_OK;
}

const char* slam_family_name(int32_t family) { return (family >= 0 && family < F_COUNT) ? kFamName[family] : nullptr; }

int slam_family_ms(SlamEngine* h, int32_t* family_out, float* ms_out, int32_t capacity, int32_t* count_out) {
  if (!h || !family_out || !ms_out || !count_out || capacity < 0) return SLAM_EINVAL;
  int32_t n = 0;
  for (const auto& mk : h->fam_marks) {
    if (n >= capacity) break;
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(h->fam_ev[mk.second + 1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->fam_ev[mk.second], h->fam_ev[mk.second + 1]);
    if (e != hipSuccess) return h->fail((int)e, "family timing events are not recorded (set time_families before the step)");
    family_out[n] = mk.first;
    ms_out[n] = ms;
    ++n;
  }
  *count_out = n;
  return SLAM_OK;
}

// fp32 gradient range <-> bf16 communication image (the data-parallel exchange in bf16: the reference's DDP reduces bf16
// gradients because its parameters are bf16, config/model/slam.yaml:9)
int slam_pack_grads_bf16(SlamEngine* h, int64_t offset, int64_t count, void* dst_bf16, slam_stream_t stream) {
  if (!h || !dst_bf16 || offset < 0 || count < 0 || offset + count > h->n_params || (offset & 3) || (count & 3)) return SLAM_EINVAL;
  if (!h->grads) return h->fail(SLAM_ESTATE, "no gradient buffer bound");
  if (count) CK(f32_to_bf16(h->grads + offset, (bf16_t*)dst_bf16, (size_t)count, (hipStream_t)stream));
  return SLAM_OK;
}
int slam_set_grad_image(SlamEngine* h, void* grads_bf16) {
  if (!h) return SLAM_EINVAL;
  h->grad_img = (bf16_t*)grads_bf16;
  return SLAM_OK;
}
int slam_unpack_grads_bf16(SlamEngine* h, int64_t offset, int64_t count, const void* src_bf16, slam_stream_t stream) {
  if (!h || !src_bf16 || offset < 0 || count < 0 || offset + count > h->n_params || (offset & 3) || (count & 3)) return SLAM_EINVAL;
  if (!h->grads) return h->fail(SLAM_ESTATE, "no gradient buffer bound");
  if (count) CK(bf16_to_f32((const bf16_t*)src_bf16, h->grads + offset, (size_t)count, (hipStream_t)stream));
  return SLAM_OK;
}

// ---- engine-side gradient exchange over RCCL (slam_allreduce_grads_async) -----------------------------------
// Replaces, for a consumer WITHOUT torch.distributed, what accelerate's DDP wrapper does for the reference
// (/root/reference config/training_args/default.yaml:18, cli/train.py:51,61: torchrun sets RANK / WORLD_SIZE, the HF Trainer
// wraps the model in DistributedDataParallel). The Python trainer keeps issuing its collectives through torch.distributed
// (slamkit_amd/trainer/dp.py): same RCCL underneath. RCCL is NOT a link-time dependency of the engine: it is looked up at the
// first slam_comm_* call (the copy a host process has already loaded - torch's - is the one dlopen returns).
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// resolved ONCE, by whichever thread gets here first (function-local static: C++11 guarantees the others wait - a
// single-process multi-GPU consumer calls slam_comm_init from one thread per rank at the same time, ncclCommInitRank blocks
// until all have joined)
const Rccl* rccl() {
  static const Rccl r = [] {
    Rccl t;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      t.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (t.lib) break;
    }
    if (t.lib) {
      t.GetUniqueId = (decltype(t.GetUniqueId))dlsym(t.lib, "ncclGetUniqueId");
      t.CommInitRank = (decltype(t.CommInitRank))dlsym(t.lib, "ncclCommInitRank");
      t.CommDestroy = (decltype(t.CommDestroy))dlsym(t.lib, "ncclCommDestroy");
      t.AllReduce = (decltype(t.AllReduce))dlsym(t.lib, "ncclAllReduce");
      t.ReduceScatter = (decltype(t.ReduceScatter))dlsym(t.lib, "ncclReduceScatter");
      t.AllGather = (decltype(t.AllGather))dlsym(t.lib, "ncclAllGather");
      t.GetErrorString = (decltype(t.GetErrorString))dlsym(t.lib, "ncclGetErrorString");
      if (!t.GetUniqueId || !t.CommInitRank || !t.CommDestroy || !t.AllReduce || !t.ReduceScatter || !t.AllGather) t.lib = nullptr;
    }
    return t;
  }();
  return r.lib ? &r : nullptr;
}
}  // namespace

int slam_comm_unique_id(void* id_out, int32_t bytes) {
  if (!id_out || bytes < (int32_t)sizeof(ncclUniqueId)) return SLAM_EINVAL;
  const Rccl* r = rccl();
  if (!r) return SLAM_EUNSUPPORTED;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return SLAM_ESTATE;
  memcpy(id_out, &id, sizeof(id));
  return SLAM_OK;
}

int slam_comm_init(SlamEngine* h, const void* id, int32_t rank, int32_t world) {
  if (!h || !id || world <= 0 || rank < 0 || rank >= world) return SLAM_EINVAL;
  if (h->comm) return h->fail(SLAM_ESTATE, "communicator already initialised");
  const Rccl* r = rccl();
  if (!r) return h->fail(SLAM_EUNSUPPORTED, "librccl.so.1 not found");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  if (!h->comm_stream) CK((int)hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));  // before the communicator: nothing to undo on failure
  ncclComm_t c = nullptr;
  const ncclResult_t e = r->CommInitRank(&c, world, uid, rank);
  if (e != ncclSuccess) return h->fail(SLAM_ESTATE, r->GetErrorString ? r->GetErrorString(e) : "ncclCommInitRank failed");
  h->comm = c;
  h->comm_world = world;
  h->comm_rank = rank;
  return SLAM_OK;
}

int slam_comm_destroy(SlamEngine* h) {
  if (!h) return SLAM_EINVAL;
  if (h->comm) {
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    const Rccl* r = rccl();
    if (r) (void)r->CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
    h->comm_world = 0;
  }
  return SLAM_OK;
}

int slam_allreduce_grads_async(SlamEngine* h, int64_t offset, int64_t count, int32_t bf16_exchange, slam_stream_t ready) {
  if (!h || offset < 0 || count < 0 || offset + count > h->n_params) return SLAM_EINVAL;
  if (!h->comm) return h->fail(SLAM_ESTATE, "slam_comm_init first");
  if (!h->grads) return h->fail(SLAM_ESTATE, "no gradient buffer bound");
  if (bf16_exchange && (!h->last_grad_img || (offset & 3) || (count & 3)))
    return h->fail(SLAM_ESTATE, "bf16 exchange: bind an image with slam_set_grad_image before the backward (ranges in multiples of 4)");
  if (!count) return SLAM_OK;
  const Rccl* r = rccl();
  // communication stream behind the producers of the range
  if (h->comm_ev_used == h->comm_ev.size()) {
    hipEvent_t e;
    CK((int)hipEventCreateWithFlags(&e, sync_event_flags()));  // SLAM_EVENT_SYSTEM_FENCE=1 covers the path whose data leaves the device
    h->comm_ev.push_back(e);
  }
  hipEvent_t ev = h->comm_ev[h->comm_ev_used++];
  CK((int)hipEventRecord(ev, (hipStream_t)ready));
  CK((int)hipStreamWaitEvent(h->comm_stream, ev, 0));
  ncclResult_t e;
  if (bf16_exchange) {
    bf16_t* img = h->last_grad_img + offset;
    e = r->AllReduce(img, img, (size_t)count, ncclBfloat16, ncclSum, (ncclComm_t)h->comm, h->comm_stream);
    if (e == ncclSuccess && h->gfinal != 2) CK(bf16_to_f32(img, h->grads + offset, (size_t)count, h->comm_stream));  // gfinal 2: read where they are
  } else {
    e = r->AllReduce(h->grads + offset, h->grads + offset, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)h->comm, h->comm_stream);
  }
  if (e != ncclSuccess) return h->fail(SLAM_ESTATE, r->GetErrorString ? r->GetErrorString(e) : "ncclAllReduce failed");
  return SLAM_OK;
}

// ---- the reduce-scatter / all-gather form of the exchange (what the Python trainer runs as ddp_algo = "rs_ag"; SURVEY.md §5
// comm row, §8e: every one of a GPU's 7 xGMI links carries one shard concurrently) for the torch-free consumer -------------------
static int comm_behind(SlamEngine* h, hipStream_t ready) {  // the communication stream continues after everything on `ready`
  if (h->comm_ev_used == h->comm_ev.size()) {
    hipEvent_t e;
    CK((int)hipEventCreateWithFlags(&e, sync_event_flags()));
    h->comm_ev.push_back(e);
  }
  hipEvent_t ev = h->comm_ev[h->comm_ev_used++];
  CK((int)hipEventRecord(ev, ready));
  CK((int)hipStreamWaitEvent(h->comm_stream, ev, 0));
  return SLAM_OK;
}

int slam_reduce_scatter_grads_async(SlamEngine* h, int64_t offset, int64_t count, int32_t bf16_exchange, slam_stream_t ready) {
  if (!h || offset < 0 || count < 0 || offset + count > h->n_params) return SLAM_EINVAL;
  if (!h->comm) return h->fail(SLAM_ESTATE, "slam_comm_init first");
  if (!h->grads) return h->fail(SLAM_ESTATE, "no gradient buffer bound");
  const int64_t w = h->comm_world, s = count / w;
  if (count % w || (s & 7) || (offset & 7)) return h->fail(SLAM_EINVAL, "reduce-scatter: count must be world x a multiple of 8 elements, offset a multiple of 8");
  if (bf16_exchange && !h->last_grad_img) return h->fail(SLAM_ESTATE, "bf16 exchange: bind an image with slam_set_grad_image before the backward");
  if (!count) return SLAM_OK;
  const Rccl* r = rccl();
  if (int rc = comm_behind(h, (hipStream_t)ready)) return rc;
  const int64_t mine = offset + (int64_t)h->comm_rank * s;
  ncclResult_t e;
  if (bf16_exchange) {
    bf16_t* img = h->last_grad_img;
    e = r->ReduceScatter(img + offset, img + mine, (size_t)s, ncclBfloat16, ncclSum, (ncclComm_t)h->comm, h->comm_stream);
    // gradients kept in bf16 only (grad_final_next = 2): the reduced shard is read where it is; else widen it into the fp32 buffer
    if (e == ncclSuccess && h->gfinal != 2) CK(bf16_to_f32(img + mine, h->grads + mine, (size_t)s, h->comm_stream));
  } else {
    e = r->ReduceScatter(h->grads + offset, h->grads + mine, (size_t)s, ncclFloat32, ncclSum, (ncclComm_t)h->comm, h->comm_stream);
  }
  if (e != ncclSuccess) return h->fail(SLAM_ESTATE, r->GetErrorString ? r->GetErrorString(e) : "ncclReduceScatter failed");
  return SLAM_OK;
}

int slam_allgather_params_async(SlamEngine* h, int64_t offset, int64_t count, slam_stream_t ready) {
  if (!h || offset < 0 || count < 0 || offset + count > h->n_params) return SLAM_EINVAL;
  if (!h->comm) return h->fail(SLAM_ESTATE, "slam_comm_init first");
  if (!h->params) return h->fail(SLAM_ESTATE, "bind params first");
  const int64_t w = h->comm_world, s = count / w;
  if (count % w || (s & 7) || (offset & 7)) return h->fail(SLAM_EINVAL, "all-gather: count must be world x a multiple of 8 elements, offset a multiple of 8");
  if (!count) return SLAM_OK;
  const Rccl* r = rccl();
  if (int rc = comm_behind(h, (hipStream_t)ready)) return rc;
  bf16_t* P = h->params;
  const ncclResult_t e = r->AllGather(P + offset + (int64_t)h->comm_rank * s, P + offset, (size_t)s, ncclBfloat16, (ncclComm_t)h->comm, h->comm_stream);
  if (e != ncclSuccess) return h->fail(SLAM_ESTATE, r->GetErrorString ? r->GetErrorString(e) : "ncclAllGather failed");
  // the next reader of the range (slam_forward, layer by layer) waits for the arrival right before its first read
  if (h->ag_ev_used == h->ag_ev.size()) {
    hipEvent_t ev;
    CK((int)hipEventCreateWithFlags(&ev, sync_event_flags()));
    h->ag_ev.push_back(ev);
  }
  hipEvent_t ev = h->ag_ev[h->ag_ev_used++];
  CK((int)hipEventRecord(ev, h->comm_stream));
  h->pwaits.push_back({offset, offset + count, ev});
  h->params_t_dirty = h->params_t != nullptr;
  return SLAM_OK;
}

int slam_comm_finish(SlamEngine* h, slam_stream_t stream) {
  if (!h) return SLAM_EINVAL;
  if (!h->comm_stream || !h->comm_ev_used) return SLAM_OK;
  if (h->comm_ev_used == h->comm_ev.size()) {
    hipEvent_t e;
    CK((int)hipEventCreateWithFlags(&e, sync_event_flags()));  // SLAM_EVENT_SYSTEM_FENCE=1 covers the path whose data leaves the device
    h->comm_ev.push_back(e);
  }
  hipEvent_t ev = h->comm_ev[h->comm_ev_used];
  CK((int)hipEventRecord(ev, h->comm_stream));
  CK((int)hipStreamWaitEvent((hipStream_t)stream, ev, 0));
  h->comm_ev_used = 0;  // the pool is reused by the next step (events are re-recorded; the waits above were already enqueued)
  return SLAM_OK;
}

int slam_param_wait_ms(SlamEngine* h, float* total_ms) {
  if (!h || !total_ms) return SLAM_EINVAL;
  float tot = 0.f;
  for (size_t i = 0; i + 1 < h->pw_used; i += 2) {
    float ms = 0.f;
    if (hipEventSynchronize(h->pw_ev[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, h->pw_ev[i], h->pw_ev[i + 1]) == hipSuccess) tot += ms;
  }
  h->pw_used = 0;
  *total_ms = tot + (float)h->pw_acc_ms;
  h->pw_acc_ms = 0.0;
  return SLAM_OK;
}

int slam_param_wait_untimed(SlamEngine* h, int64_t* n) {
  if (!h || !n) return SLAM_EINVAL;
  *n = h->pw_untimed;
  h->pw_untimed = 0;
  return SLAM_OK;
}

int slam_join(SlamEngine* h, slam_stream_t stream) {
  if (!h) return SLAM_EINVAL;
  CK(join_optimizer(h, (hipStream_t)stream));
  CK(join_params(h, (hipStream_t)stream));
  return SLAM_OK;
}

int slam_zero_grads(SlamEngine* h, slam_stream_t stream) {
  if (!h || !h->grads) return SLAM_EINVAL;
  CK(join_optimizer(h, (hipStream_t)stream));
  CK((int)hipMemsetAsync(h->grads, 0, (size_t)h->n_params * sizeof(float), (hipStream_t)stream));
  return SLAM_OK;
}

int slam_cast_params(SlamEngine* h, const float* master, slam_stream_t stream) {
  if (!h || !master || !h->params) return SLAM_EINVAL;
  CK(join_optimizer(h, (hipStream_t)stream));
  CK(f32_to_bf16(master, h->params, (size_t)h->n_params, (hipStream_t)stream));
  return slam_refresh_transposed(h, stream);
}

// ---- single-op entry points ------------------------------------------------------------------
int slam_op_gemm_nt(const void* X, const void* W, void* Y, const void* bias, const void* resid, int M, int N, int K,
                    int use_glds, slam_stream_t s) {
  GemmTune t = *gemm_default_tune();  // the process default with the staging mode of this call
  t.glds = use_glds != 0;
  GemmTuneScope scope(&t);
  return gemm_nt((const bf16_t*)X, (const bf16_t*)W, (bf16_t*)Y, (const bf16_t*)bias, (const bf16_t*)resid, M, N, K, (hipStream_t)s);
}
int slam_op_gemm_nt_swiglu(const void* X, const void* W, void* Y, void* act, int M, int N, int K, slam_stream_t s) {
  return gemm_nt_swiglu((const bf16_t*)X, (const bf16_t*)W, (bf16_t*)Y, (bf16_t*)act, M, N, K, (hipStream_t)s);
}
int slam_op_gemm_nt_dswiglu(const void* dY, const void* Wt, void* gu, int M, int I, int H, slam_stream_t s) {
  return gemm_nt_dswiglu((const bf16_t*)dY, (const bf16_t*)Wt, (bf16_t*)gu, M, I, H, (hipStream_t)s);
}
int slam_op_gemm_nn(const void* dY, const void* W, void* dX, const void* resid, int M, int N, int K, slam_stream_t s) {
  return gemm_nn((const bf16_t*)dY, (const bf16_t*)W, (bf16_t*)dX, (const bf16_t*)resid, M, N, K, (hipStream_t)s);
}
size_t slam_op_gemm_tn_workspace(int M, int N, int K) { return gemm_tn_workspace_bytes(M, N, K); }
int slam_op_gemm_tn(const void* dY, const void* X, float* dW, int accumulate, int M, int N, int K, float* ws,
                    slam_stream_t s) {
  // capacity = what slam_op_gemm_tn_workspace(M, N, K) promises (the caller's contract; recomputing the bound here would put
  // a host-side planning sweep into every launch)
  static int cm = 0, cn = 0, ck = 0;
  static size_t cap = 0;
  if (cm != M || cn != N || ck != K) { cap = gemm_tn_workspace_bytes(M, N, K); cm = M; cn = N; ck = K; }
  return gemm_tn((const bf16_t*)dY, (const bf16_t*)X, dW, accumulate, M, N, K, N, K, ws, cap, (hipStream_t)s);
}
int slam_op_gemm_tn_image(const void* dY, const void* X, float* dW, void* dW_bf16, int accumulate, int M, int N, int K, float* ws,
                          int background, slam_stream_t s) {
  const size_t cap = gemm_tn_workspace_bytes(M, N, K);
  return gemm_tn((const bf16_t*)dY, (const bf16_t*)X, dW, accumulate, M, N, K, N, K, ws, cap, (hipStream_t)s, background, (bf16_t*)dW_bf16);
}
int slam_op_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, slam_stream_t s) {
  return rmsnorm_fwd((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, M, H, eps, (hipStream_t)s);
}
size_t slam_op_rmsnorm_bwd_workspace(int M, int H) { return (size_t)rmsnorm_bwd_blocks(M) * H * sizeof(float); }
int slam_op_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                        float* dw, float* ws, int M, int H, slam_stream_t s) {
  return rmsnorm_bwd((const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw,
                     0, ws, M, H, (hipStream_t)s);
}
int slam_op_rope(void* qkv, int ld, int M, int T, int n_rot_heads, int head_dim, const int64_t* position_ids, float theta,
                 int backward, float* cs_ws, slam_stream_t s) {
  const size_t half = (size_t)head_dim / 2;
  int r = rope_table(position_ids, M, T, head_dim, theta, cs_ws, cs_ws + (size_t)M * half, nullptr, nullptr, 1.f, (hipStream_t)s);
  if (r) return r;
  return rope_apply((bf16_t*)qkv, ld, M, n_rot_heads, head_dim, cs_ws, cs_ws + (size_t)M * half, backward, (hipStream_t)s);
}
int slam_op_swiglu_fwd(const void* gu, void* act, int M, int I, slam_stream_t s) {
  return swiglu_fwd((const bf16_t*)gu, (bf16_t*)act, M, I, I, (hipStream_t)s);
}
int slam_op_swiglu_bwd(void* gu, const void* dact, int M, int I, slam_stream_t s) {
  return swiglu_bwd((bf16_t*)gu, (const bf16_t*)dact, M, I, I, (hipStream_t)s);
}
int slam_op_attn_fwd(const void* qkv, void* o, float* lse2, const int32_t* seg_start, int M, int nH, int nKV,
                     int head_dim, slam_stream_t s) {
  // the forward runs in the engine's heaviest-first block order; the op entry keeps a process-lifetime plan buffer for it
  static int* plan = nullptr;
  static size_t cap = 0;
  const size_t need = attn_plan_ints(M);
  if (need > cap) {
    if (plan) (void)hipFree(plan);
    if (hipMalloc(&plan, need * sizeof(int)) != hipSuccess) { plan = nullptr; cap = 0; return (int)hipErrorOutOfMemory; }
    cap = need;
  }
  const AttnTune tune = attn_default_tune();
  int r = attn_plan(seg_start, nullptr, M, head_dim, tune, plan, (hipStream_t)s);
  if (r) return r;
  return attn_fwd((const bf16_t*)qkv, (bf16_t*)o, lse2, seg_start, plan, tune, M, nH, nKV, head_dim, (hipStream_t)s);
}
size_t slam_op_attn_bwd_workspace(int M, int nH, int head_dim) {
  // the ABI call has no KV-head count: sized for nKV = nH (plain multi-head attention), the largest case
  return attn_bwd_workspace_bytes(M, nH, head_dim) + (size_t)2 * M * nH * sizeof(float) + attn_plan_ints(M) * sizeof(int) + 64;
}
int slam_op_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, float* ws,
                     const int32_t* seg_start, const int32_t* seg_end, int M, int nH, int nKV, int head_dim,
                     slam_stream_t s) {
  float* ndsum = ws;
  float* nlse = ws + (size_t)M * nH;
  float* part = ws + (size_t)2 * M * nH;
  int* plan = reinterpret_cast<int*>(part + attn_bwd_workspace_bytes(M, nH, head_dim) / sizeof(float));
  const AttnTune tune = attn_default_tune();
  int r = attn_plan(seg_start, seg_end, M, head_dim, tune, plan, (hipStream_t)s);
  if (r) return r;
  return attn_bwd((const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse2, ndsum, nlse, (bf16_t*)dqkv, part, seg_start,
                  seg_end, plan, tune, nullptr, nullptr, M, nH, nKV, head_dim, (hipStream_t)s);
}
int slam_op_cross_entropy(const void* logits, const int64_t* labels, double num_items, void* dlogits, float* row_loss,
                          float* scratch2, int B, int T, int Vp, int V, slam_stream_t s) {
  return cross_entropy((const bf16_t*)logits, labels, num_items, (bf16_t*)dlogits, row_loss, scratch2, scratch2 + 1, B,
                       T, Vp, V, nullptr, (hipStream_t)s);
}
size_t slam_op_embed_bwd_workspace(int M, int Vp) { return embed_bwd_workspace_ints(M, Vp) * sizeof(int); }
int slam_op_embed_bwd(const int64_t* ids, const void* dh, float* dE, int M, int H, int Vp, int V, int pad_id, void* ws,
                      slam_stream_t s) {
  return embed_bwd(ids, (const bf16_t*)dh, dE, M, H, Vp, V, pad_id, (int*)ws, (hipStream_t)s);
}

}  // extern "C"
