/*
 * slam_engine.h - C ABI of libslam_engine.so, the gfx950 (MI355X) engine behind slamkit's
 * cli/train.py hot path.
 *
 * The reference has no FFI on this path: the boundary is the Python plugin surface
 *   tlm_factory(cfg.model) -> TokenLM            /root/reference slamkit/model/token_lm.py:30-43
 *   UnitLM.forward(input_ids, attention_mask, position_ids, labels, num_items_in_batch)
 *                                                 slamkit/model/unit_lm.py:135-182
 *   compute_loss(logits, labels, num_items_in_batch)   slamkit/model/unit_lm.py:13-29
 *   UnitLM.log_likelihood                          slamkit/model/unit_lm.py:184-194
 *   SLAMTrainer.training_step + HF Trainer step    slamkit/trainer/slam_trainer.py:59-71
 * Each entry point below names the reference call it replaces. All device pointers are BORROWED
 * (PyTorch-ROCm tensor.data_ptr()); the engine never frees them and never synchronises the host:
 * every call only enqueues work on the hipStream_t it is given. Return value 0 = ok, negative =
 * SLAM_E*, positive = hipError_t; slam_last_error() gives text. No exceptions cross the ABI.
 * One engine per process/GPU; calls on one engine must be serialised by the caller.
 */
#ifndef SLAM_ENGINE_H
#define SLAM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_OK 0
#define SLAM_EINVAL (-1)      /* bad argument / unsupported shape */
#define SLAM_ESTATE (-2)      /* call order (e.g. backward without forward, unbound buffers) */
#define SLAM_ENOMEM (-3)      /* bound workspace too small */
#define SLAM_EUNSUPPORTED (-4) /* an optional run-time dependency is missing (slam_comm_*: librccl) */

typedef struct SlamEngine SlamEngine;
typedef void* slam_stream_t; /* hipStream_t */

/* Qwen2-shaped decoder description (UnitLMConfig.base_config, unit_lm.py:32-79; Slam-358M values
 * from config/model/slam.yaml + Qwen2.5-0.5B). */
typedef struct SlamModelDesc {
  int32_t n_layers;      /* 24  */
  int32_t hidden;        /* 896 */
  int32_t n_heads;       /* 14  */
  int32_t n_kv_heads;    /* 2   */
  int32_t head_dim;      /* 64 or 128 */
  int32_t intermediate;  /* 4864 */
  int32_t vocab;         /* 502; embedding rows are padded to 512, or to a multiple of 256 beyond 512 */
  int32_t pad_token_id;  /* 0: nn.Embedding(padding_idx) gather-gradient suppression; -1 = none */
  float rms_eps;         /* 1e-6 */
  float rope_theta;      /* 10000 */
} SlamModelDesc;

typedef struct SlamTensorInfo {
  char name[64];      /* "embed", "layers.3.wqkv", "layers.3.bqkv", "layers.3.wo", "layers.3.ln1",
                         "layers.3.ln2", "layers.3.wgu", "layers.3.wd", "norm".
                         wqkv rows = q | k | v; wgu rows = blocks of 32 gate_proj rows followed by the
                         matching 32 up_proj rows (row 64b+j = gate[32b+j], row 64b+32+j = up[32b+j]) */
  int64_t offset;     /* element offset in the flat parameter / gradient buffers */
  int64_t rows, cols; /* row-major [rows][cols] (cols = 1 for vectors) */
} SlamTensorInfo;

/* Called from slam_backward (host side, after the producing kernels were enqueued) when the
 * gradient range [offset, offset+count) of the flat fp32 gradient buffer is final. Used by the
 * data-parallel reducer to launch RCCL all-reduces overlapped with the rest of backward
 * (replaces torch DDP's reducer hooks). */
typedef void (*slam_bucket_cb)(void* user, int64_t offset, int64_t count);

/* ---- lifetime -------------------------------------------------------------------------------*/
int slam_engine_create(const SlamModelDesc* desc, SlamEngine** out); /* UnitLM.__init__ :91-112 */
void slam_engine_destroy(SlamEngine* h);
const char* slam_last_error(SlamEngine* h);
const char* slam_version(void);

/* ---- parameter layout -----------------------------------------------------------------------*/
int64_t slam_param_count(SlamEngine* h);   /* elements in the flat buffers (padded vocab rows incl.) */
int32_t slam_tensor_count(SlamEngine* h);
int slam_tensor_info(SlamEngine* h, int32_t index, SlamTensorInfo* out);
/* params: bf16 [slam_param_count], grads: fp32 [slam_param_count] (model.parameters() / .grad) */
int slam_bind_params(SlamEngine* h, void* params_bf16, float* grads_f32);
/* Optional bf16 [slam_param_count] buffer that receives every weight matrix TRANSPOSED (same
 * offsets): with it the dgrad GEMMs dX = dY W run as the fast contraction-contiguous form
 * dY (W^T)^T. Refreshed by slam_adamw_step / slam_cast_params / slam_refresh_transposed. */
int slam_bind_params_t(SlamEngine* h, void* params_t_bf16);
int slam_refresh_transposed(SlamEngine* h, slam_stream_t stream);

/* ---- workspace ------------------------------------------------------------------------------*/
size_t slam_workspace_bytes(SlamEngine* h, int64_t max_tokens);
int slam_bind_workspace(SlamEngine* h, void* ws, size_t bytes, int64_t max_tokens);

/* ---- forward / loss: UnitLM.forward + compute_loss (unit_lm.py:13-29,135-182) ----------------
 * ids/labels/position_ids: int64 [B*T] device (labels, position_ids nullable).
 * seg_start/seg_end: int32 [B*T] device, nullable -> dense rows of length T. For packed batches
 *   ([1, sum T] from DataCollatorWithFlattening) they give each token's sequence bounds.
 * num_items > 0 -> loss = sum / num_items (reduction "sum"), else mean over valid targets.
 * loss_out: fp32 [1] device (nullable when labels is NULL); logits_out: bf16 [B*T*vocab] device,
 * nullable. */
int slam_forward(SlamEngine* h, const int64_t* ids, const int64_t* labels, const int64_t* position_ids,
                 const int32_t* seg_start, const int32_t* seg_end, int32_t B, int32_t T, double num_items,
                 float* loss_out, void* logits_out, slam_stream_t stream);

/* loss.backward(): accumulates d(loss*grad_scale)/dparam into the bound fp32 gradient buffer.
 * bucket_layers = decoder layers per gradient bucket for the callback (<=0: one bucket). */
int slam_backward(SlamEngine* h, float grad_scale, int32_t bucket_layers, slam_bucket_cb cb, void* user,
                  slam_stream_t stream);
/* ---- engine-side gradient exchange (RCCL over xGMI) for a consumer without torch.distributed -----------------------------
 * What DistributedDataParallel does for the reference (/root/reference config/training_args/default.yaml:18, cli/train.py:51,61;
 * SURVEY.md §8a T10, §8b `slam_allreduce_grads_async`). One communicator per engine on the engine's device; RCCL is looked up at the
 * first call (dlopen of librccl.so.1 - the copy the process already has is reused), SLAM_EUNSUPPORTED when it is absent.
 *   rank 0: slam_comm_unique_id(id, 128) -> ship the 128 bytes to every rank -> each rank: slam_comm_init(h, id, rank, world)
 *   per step: slam_backward(h, scale, bucket_layers, cb, ...) with a callback that calls
 *             slam_allreduce_grads_async(h, offset, count, bf16, slam_bucket_stream(h) ? slam_bucket_stream(h) : stream)
 *             -> slam_comm_finish(h, stream) -> slam_grad_norm / slam_adamw_step on `stream`.
 * slam_allreduce_grads_async: grads[offset, offset + count) summed over the ranks (in place) on the engine's communication stream,
 * ordered after everything enqueued so far on `ready`; it returns at once. bf16_exchange = 1: the bf16 image bound by
 * slam_set_grad_image before that backward crosses the wire (half the bytes) and is widened back into the fp32 buffer.
 * slam_comm_finish: `stream` waits for every exchange issued since the last finish. */
#define SLAM_COMM_ID_BYTES 128
int slam_comm_unique_id(void* id_out, int32_t bytes);
int slam_comm_init(SlamEngine* h, const void* id, int32_t rank, int32_t world);
int slam_comm_destroy(SlamEngine* h);
int slam_allreduce_grads_async(SlamEngine* h, int64_t offset, int64_t count, int32_t bf16_exchange, slam_stream_t ready);
int slam_comm_finish(SlamEngine* h, slam_stream_t stream);
/* The reduce-scatter / all-gather form of the same exchange (what slamkit_amd's trainer runs as ddp_algo = "rs_ag"; SURVEY.md
 * §5 comm row, §8e): per bucket the ranks reduce-scatter the gradients - rank r ends up with the summed shard
 * [offset + r s, offset + (r + 1) s), s = count / world (count = world x a multiple of 8) - the optimizer runs on the owned shards
 * (slam_grad_sumsq_chunks + one all-reduce of the chunk sums, slam_adamw_range*), and the updated bf16 PARAMETERS are all-gathered:
 *   callback: slam_reduce_scatter_grads_async(h, offset, count, bf16, ready)      -> slam_comm_finish(h, stream)
 *   update of the owned shards on `stream`, then per bucket, lowest offsets first:
 *             slam_allgather_params_async(h, offset, count, stream)
 * The gather runs on the communication stream under the next forward, which waits for each bucket right before its first read
 * (the engine registers the wait itself - slam_add_param_wait is for consumers with their own communicator). */
int slam_reduce_scatter_grads_async(SlamEngine* h, int64_t offset, int64_t count, int32_t bf16_exchange, slam_stream_t ready);
int slam_allgather_params_async(SlamEngine* h, int64_t offset, int64_t count, slam_stream_t ready);
/* Valid inside a slam_bucket_cb call: the stream on which the reported range is complete - the consumer records its
 * "bucket ready" event THERE. NULL = the stream passed to slam_backward. With the weight-gradient stream on, the
 * intermediate buckets are complete on that engine-owned stream (which has also been ordered after the norm / bias
 * kernels of the range on `stream`), so `stream` itself never waits for the weight gradients at a bucket boundary. */
slam_stream_t slam_bucket_stream(SlamEngine* h);

/* Per-sequence log-likelihood sums of the last forward (UnitLM.log_likelihood :184-194 /
 * calc_nll, slamkit/utils/calculation_utils.py:5-29): ll_out, cnt_out fp32 [B] device. */
/* Modality-restricted scoring (UnitLM.log_likelihood(ignore_tokens=...), unit_lm.py:185-188): columns whose byte
 * in `mask` (device pointer to slam_padded_vocab(h) bytes, borrowed until reset with NULL) is non-zero are treated
 * as -inf by the loss of every following slam_forward; a target inside the mask gives an infinite row loss. */
int slam_set_logit_mask(SlamEngine* h, const uint8_t* mask);
int32_t slam_padded_vocab(SlamEngine* h);
int slam_seq_loglik(SlamEngine* h, const int64_t* labels, int32_t B, int32_t T, float* ll_out, float* cnt_out,
                    slam_stream_t stream);

/* Sequence-level objectives on top of the token log-likelihoods (preference optimisation, TRL DPOTrainer
 * behind /root/reference cli/preference_alignment_train.py:56-65): after a forward with labels and
 * num_items = 1 (so d loss/d logits holds softmax - onehot per valid token), multiply the rows of sequence b
 * by seq_coef[b] (fp32 [B] device) - e.g. +-beta*sigmoid(-x)/n for the sigmoid DPO loss - then slam_backward. */
int slam_scale_loss_rows(SlamEngine* h, const float* seq_coef, int32_t B, int32_t T, slam_stream_t stream);

/* ---- optimizer step: HF Trainer clip_grad_norm_ + torch AdamW --------------
 * norm_out: fp32 [2] device = {global grad norm, clip coefficient}.
 * Where the gradients are read from follows the last slam_backward: the fp32 buffer of slam_bind_params, or - after a
 * backward that ran under slam_set_option(h, "grad_final_next", 2) - the bf16 buffer given to slam_set_grad_image, which then
 * holds the ONLY copy of the step's final gradient values (the reference's own gradient precision: bf16 parameters have
 * bf16 .grad, /root/reference config/model/slam.yaml:9). After "grad_final_next" >= 1 slam_grad_norm adds the per-block sums
 * of squares that backward's final-value stores emitted, in launch / block order (the same bits every run), instead of
 * reading the buffer again. */
int slam_grad_norm(SlamEngine* h, float max_norm, float* norm_out, slam_stream_t stream);
int slam_adamw_step(SlamEngine* h, float* master_f32, float* exp_avg, float* exp_avg_sq, const float* norm_out,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step,
                    int32_t zero_grad, slam_stream_t stream);
/* The Slam recipe's own optimizer precision (/root/reference config/model/slam.yaml:9 `torch_dtype: bfloat16`: bf16
 * parameters and bf16 Adam moments, torch.optim.AdamW(fused) semantics: fp32 arithmetic per element from the stored bf16
 * values, one rounding on the way back, no fp32 master copy). Updates the BOUND bf16 parameter buffer in place and
 * refreshes the transposed images; exp_avg / exp_avg_sq: bf16 [slam_param_count]. 16 B/param instead of 30. */
int slam_adamw_step_bf16(SlamEngine* h, void* exp_avg_bf16, void* exp_avg_sq_bf16, const float* norm_out, double lr,
                         double beta1, double beta2, double eps, double weight_decay, int32_t step, int32_t zero_grad,
                         slam_stream_t stream);
/* The middle precision: fp32 master weights, Adam moments STORED in bf16 (fp32 arithmetic per element, one rounding on the
 * way back): 22 B/param instead of 30. With transposed weight images bound, all three forms write those images from the
 * optimizer kernel itself (64 x 64 tiles through LDS) - there is no separate transpose pass after the update. */
int slam_adamw_step_bf16_moments(SlamEngine* h, float* master_f32, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                                 const float* norm_out, double lr, double beta1, double beta2, double eps, double weight_decay,
                                 int32_t step, int32_t zero_grad, slam_stream_t stream);
/* ---- sharded optimizer step: the data-parallel "rs_ag" exchange (replaces torch DDP's all-reduce of
 * every gradient followed by N identical optimizer steps, /root/reference config/training_args/default.yaml:18 +
 * site-packages transformers/trainer.py) -------------------------------------------------------------------------------
 * Per gradient bucket the ranks reduce-scatter the gradients, each rank updates the 1/N shard it owns, and the bf16
 * parameters are all-gathered while the next forward already runs. The global gradient norm is defined on fixed chunks of
 * slam_grad_chunk_elems() consecutive elements of the flat buffer: slam_grad_sumsq_chunks fills chunk_sums[k] (fp32
 * [ceil(slam_param_count / chunk)], device) for the chunk-aligned range it is given - a rank fills the chunks of its own
 * shards and leaves zeros elsewhere, the arrays are summed over the ranks (exact: disjoint support), and
 * slam_grad_norm_from_chunks finishes {norm, clip coefficient} exactly as slam_grad_norm does for the whole buffer (which
 * computes the same chunk sums itself): the sharded and the replicated step clip with bit-identical coefficients. */
int64_t slam_grad_chunk_elems(void);
int slam_grad_sumsq_chunks(SlamEngine* h, int64_t offset, int64_t count, float* chunk_sums, slam_stream_t stream);
int slam_grad_norm_from_chunks(SlamEngine* h, const float* chunk_sums, float max_norm, float* norm_out, slam_stream_t stream);
/* slam_adamw_step on elements [offset, offset + count) only (multiples of 4; 8 for the bf16-state form): master / exp_avg /
 * exp_avg_sq point at THE RANGE'S first element (compact per-shard storage or base + offset of full-size buffers). Does not
 * refresh the transposed weight images: the next slam_backward does, after every pending parameter write has landed. */
int slam_adamw_range(SlamEngine* h, int64_t offset, int64_t count, float* master_f32, float* exp_avg, float* exp_avg_sq,
                     const float* norm_out, double lr, double beta1, double beta2, double eps, double weight_decay,
                     int32_t step, int32_t zero_grad, slam_stream_t stream);
int slam_adamw_range_bf16_moments(SlamEngine* h, int64_t offset, int64_t count, float* master_f32, void* exp_avg_bf16,
                                  void* exp_avg_sq_bf16, const float* norm_out, double lr, double beta1, double beta2, double eps,
                                  double weight_decay, int32_t step, int32_t zero_grad, slam_stream_t stream);
int slam_adamw_range_bf16(SlamEngine* h, int64_t offset, int64_t count, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                          const float* norm_out, double lr, double beta1, double beta2, double eps, double weight_decay,
                          int32_t step, int32_t zero_grad, slam_stream_t stream);
/* Another stream (the all-gather of a parameter bucket) is still writing the bound bf16 parameters in [offset, offset +
 * count); `event` (hipEvent_t, owned by the caller, alive until the next forward was enqueued) is recorded behind that
 * write. The next slam_forward waits for it right before its first read of the range - layer by layer, so the gather of
 * the later layers runs under the first layers' kernels; every other entry point that touches parameters waits for all
 * of them first. slam_param_wait_ms: total stall of the caller's stream in those waits since the last query (host-
 * synchronising: logging only; the waits are bracketed by timing events only while slam_set_option(h, "time_param_waits", 1)).
 * After a ranged update the transposed weight images are rebuilt at the start of the next slam_backward. */
int slam_add_param_wait(SlamEngine* h, int64_t offset, int64_t count, void* event);
int slam_param_wait_ms(SlamEngine* h, float* total_ms);
/* Waits since the last call that could NOT be bracketed by timing events (the event pool - 2048 pairs, completed pairs are
 * folded into the running total and reused - was full of pairs still in flight): 0 means slam_param_wait_ms is complete. */
int slam_param_wait_untimed(SlamEngine* h, int64_t* n);
/* Measurement hook of bench.py's `roofline`: with slam_set_option(h, "time_gateup", 1) every forward brackets the gate|up
 * projection launch of each layer (the dominant kernel: fused SwiGLU GEMM, 2 M (2I) H flop) with two timing events on the
 * caller's stream; slam_gateup_launch_ms writes the n_layers durations of the last forward (ms; host-synchronising).
 * It times the launch where it runs - inside the step, between its neighbours - which is what rocprofv3 reports for the
 * same launches. No reference counterpart (the reference has no kernel-level timing). */
int slam_gateup_launch_ms(SlamEngine* h, float* ms_out, int32_t n);
/* The same hook for EVERY launch family of the step: with slam_set_option(h, "time_families", 1) slam_forward and
 * slam_backward bracket each launch (projection / attention / norm / loss kernels, the dgrad chain, and the weight-gradient
 * GEMMs on the engine's side stream(s)) with a timing-event pair on the stream the launch goes to. slam_family_ms writes up
 * to `capacity` (family id, ms) records of the last forward + backward in launch order and their number (host-synchronising);
 * slam_family_name maps an id to its name ("gateup_fwd", "wgu_wgrad", ... ; NULL past the last id). A record is the time
 * between the launch's stream reaching it and the launch completing - beside whatever the other stream runs - i.e. the
 * in-step duration a kernel trace reports, not a stand-alone time. Two more packets per launch: measurement steps only.
 * No reference counterpart. */
int slam_family_ms(SlamEngine* h, int32_t* family_out, float* ms_out, int32_t capacity, int32_t* count_out);
const char* slam_family_name(int32_t family);
/* Data-parallel gradient exchange in bf16 (replaces the dtype conversion torch DDP never needs because the reference's
 * gradients ARE bf16: /root/reference config/model/slam.yaml:9 with config/training_args/default.yaml:18): pack
 * grads[offset, offset + count) (fp32) into a bf16 communication buffer (round-to-nearest-even) / widen a reduced bf16 range
 * back into the fp32 gradient buffer. offset and count are multiples of 4; dst / src point at the range's first element. */
int slam_pack_grads_bf16(SlamEngine* h, int64_t offset, int64_t count, void* dst_bf16, slam_stream_t stream);
/* ... or no pack pass at all: grads_bf16 = a bf16 buffer of slam_param_count() elements (NULL = off). The NEXT slam_backward
 * stores every FINAL gradient value there as well, rounded to nearest even, from the kernels that store the fp32 value (weight-
 * gradient tile epilogues, slab reduces, norm / bias finish) - bit-identical to slam_pack_grads_bf16 over the same range, for
 * 2 B/param of extra stores instead of a 6 B/param pass beside a backward that has no idle bandwidth. A range reported through
 * slam_bucket_cb is complete in the image as well. Consumed by that one backward (the last micro-batch of a step). */
int slam_set_grad_image(SlamEngine* h, void* grads_bf16);
int slam_unpack_grads_bf16(SlamEngine* h, int64_t offset, int64_t count, const void* src_bf16, slam_stream_t stream);
/* With slam_set_option(h, "overlap_adamw", 1), slam_adamw_step returns after forking the update onto an engine-owned side
 * stream in per-layer chunks; the next slam_forward waits for chunk l right before layer l and every other entry point
 * joins first. slam_join makes `stream` wait for a pending update before the caller touches the parameter, gradient or
 * optimizer buffers itself (checkpointing, logging). */
int slam_join(SlamEngine* h, slam_stream_t stream);
int slam_zero_grads(SlamEngine* h, slam_stream_t stream);
int slam_cast_params(SlamEngine* h, const float* master_f32, slam_stream_t stream); /* fp32 -> bound bf16 */

/* ---- tuning knobs ---------------------------------------------------------------------------*/
/* Tuning / mode switches. "attn_jq" / "attn_kw" (1 or 2: 16-row fragments per wave in the attention dQ / dK-dV kernels), "attn_nch" (1..4 query-range
 * chunks per key tile), "attn_prio" (wave priority by block length); with h = NULL they set the process default used by the
 * single-op entry points. "bwd_wgrad_cus" = N > 0: the weight-gradient stream is created with a CU mask of N CUs (a BLOCKING
 * stream: run the step on a non-default stream then). "grad_overwrite_next" = 1: the next slam_backward stores the gradients instead of adding to
 * them (first micro-batch of an optimizer step; no zeroing pass needed), then resets itself. "grad_final_next" = 1 | 2: the next
 * slam_backward is the LAST of its optimizer step (HF Trainer: the micro-batch on which `sync_gradients` is true) - every kernel
 * that stores a final gradient value also emits its block's sum of squares for slam_grad_norm; with 2 the final values are
 * stored ONLY as bf16 into the slam_set_grad_image buffer (earlier micro-batches keep accumulating in fp32) and slam_grad_norm /
 * slam_adamw_step* read them there; resets itself. The partial sums are not emitted by a backward that reports buckets to a callback
 * (data parallel: the clip needs the norm of the EXCHANGED gradients - chunk sums after the exchange) nor under
 * "grad_norm_partials" = 0 (slam_grad_norm then always runs the chunked pass, over whichever buffer holds the gradients). "bwd_wgrad_stream" (default
 * 1): slam_backward enqueues the weight-gradient GEMMs on an engine-owned second stream, ordered by events against the
 * dgrad chain on `stream`; `stream` is joined with it before slam_backward returns control of the gradient buffer (every
 * reported bucket range, and the end of the call). "overlap_adamw", "fuse_swiglu", "fuse_dswiglu", "gemm_256",
 * "gemm_256_dswiglu", "gemm_256_persist", "gemm_tn224", "gemm_tn_balanced", "gemm_group_rows" select kernels. */
int slam_set_option(SlamEngine* h, const char* key, int64_t value);

/* ---- single-op entry points (parity tests call each kernel through the ABI) -------------------*/
int slam_op_gemm_nt(const void* X, const void* W, void* Y, const void* bias, const void* resid, int M, int N, int K,
                    int use_glds, slam_stream_t s);
/* gate|up projection with the SwiGLU product fused into the epilogue (W rows in 32-row gate/up blocks):
 * Y[M,N] = X W^T and act[M,N/2] = silu(gate) * up. The dominant kernel of the step (bench.py roofline). */
int slam_op_gemm_nt_swiglu(const void* X, const void* W, void* Y, void* act, int M, int N, int K, slam_stream_t s);
/* down-projection dgrad with the SwiGLU backward fused into the epilogue: d(act)[M,I] = dY[M,H] Wt[I,H]^T never leaves the
 * registers; gu[M,2I] (gate|up pre-activations, 32-column gate/up blocks) is rewritten in place with d(gate|up). */
int slam_op_gemm_nt_dswiglu(const void* dY, const void* Wt, void* gu, int M, int I, int H, slam_stream_t s);
int slam_op_gemm_nn(const void* dY, const void* W, void* dX, const void* resid, int M, int N, int K, slam_stream_t s);
size_t slam_op_gemm_tn_workspace(int M, int N, int K);
int slam_op_gemm_tn(const void* dY, const void* X, float* dW, int accumulate, int M, int N, int K, float* ws,
                    slam_stream_t s);
/* the same weight-gradient GEMM, also writing the bf16 image of every final dW value (slam_set_grad_image's mechanism);
 * background != 0 selects the plans slam_backward uses on its weight-gradient stream */
int slam_op_gemm_tn_image(const void* dY, const void* X, float* dW, void* dW_bf16, int accumulate, int M, int N, int K,
                          float* ws, int background, slam_stream_t s);
int slam_op_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, slam_stream_t s);
size_t slam_op_rmsnorm_bwd_workspace(int M, int H);
int slam_op_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                        float* dw, float* ws, int M, int H, slam_stream_t s);
int slam_op_rope(void* qkv, int ld, int M, int T, int n_rot_heads, int head_dim, const int64_t* position_ids, float theta,
                 int backward, float* cos_sin_ws /* 2*M*(head_dim/2) floats */, slam_stream_t s);
int slam_op_swiglu_fwd(const void* gu, void* act, int M, int I, slam_stream_t s);
int slam_op_swiglu_bwd(void* gu_inout, const void* dact, int M, int I, slam_stream_t s);
/* attention ops: the QUERY columns of qkv are expected PRE-SCALED by head_dim^-0.5 * log2(e) (the engine's QKV projection
 * folds that factor into the queries' RoPE rotation: one rounding); dqkv's query columns come back as the gradient of
 * the UNSCALED queries. */
int slam_op_attn_fwd(const void* qkv, void* o, float* lse2, const int32_t* seg_start, int M, int nH, int nKV,
                     int head_dim, slam_stream_t s);
size_t slam_op_attn_bwd_workspace(int M, int nH, int head_dim);
int slam_op_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse2, void* dqkv, float* ws,
                     const int32_t* seg_start, const int32_t* seg_end, int M, int nH, int nKV, int head_dim,
                     slam_stream_t s);
int slam_op_cross_entropy(const void* logits /* bf16 [B*T][Vp] */, const int64_t* labels, double num_items,
                          void* dlogits, float* row_loss, float* scratch2 /* {denom, loss} */, int B, int T,
                          int Vp /* padded row length: 512, or a multiple of 8 beyond */, int V, slam_stream_t s);
/* gather-side embedding gradient for vocabularies beyond 512: dE[ids[m]] += dh[m], token order (deterministic) */
size_t slam_op_embed_bwd_workspace(int M, int Vp);
int slam_op_embed_bwd(const int64_t* ids, const void* dh /* bf16 [M][H] */, float* dE /* fp32 [Vp][H] */, int M, int H,
                      int Vp, int V, int pad_id, void* ws, slam_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_ENGINE_H */
