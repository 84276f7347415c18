/* A C consumer of libslam_engine.so WITHOUT torch: one data-parallel optimizer step through the C ABI (include/slam_engine.h).
 * Shown for the reference maintainer who binds the engine from something other than Python; compiled (not run) by
 * tests/test_abi_exports.py::test_c_consumer_compiles_against_the_header, so the header stays plain C and the sequence below
 * stays in step with it. Device memory, streams and the way the 128-byte RCCL id travels to the other ranks are the host's
 * business (hipMalloc / hipStreamCreate / MPI_Bcast ...): this file only shows the engine calls and their order.
 *
 *   cc -I include -c tools/examples/c_dp_consumer.c
 */
#include <stddef.h>
#include <stdint.h>

#include "slam_engine.h"

struct step_ctx {
  SlamEngine* h;
  slam_stream_t stream; /* the stream slam_backward was enqueued on */
  int bf16_exchange;
  int rc;
};

/* slam_bucket_cb: gradients [offset, offset + count) are final - exchange them while backward goes on */
static void on_bucket(void* user, int64_t offset, int64_t count) {
  struct step_ctx* c = (struct step_ctx*)user;
  slam_stream_t ready = slam_bucket_stream(c->h); /* the engine's weight-gradient stream for intermediate buckets */
  int rc = slam_allreduce_grads_async(c->h, offset, count, c->bf16_exchange, ready ? ready : c->stream);
  if (rc != SLAM_OK && c->rc == SLAM_OK) c->rc = rc;
}

/* one optimizer step on this rank's micro-batch; every pointer is a device pointer owned by the caller */
int dp_step(SlamEngine* h, slam_stream_t stream, const int64_t* ids, const int64_t* labels, int32_t B, int32_t T,
            double global_num_items, float* loss_dev, void* grad_image_bf16 /* n_params bf16, or NULL for an fp32 exchange */,
            void* exp_avg_bf16, void* exp_avg_sq_bf16, float* norm_out_dev, double lr, int32_t step) {
  struct step_ctx c = {h, stream, grad_image_bf16 != NULL, SLAM_OK};
  int rc = slam_forward(h, ids, labels, NULL, NULL, NULL, B, T, global_num_items, loss_dev, NULL, stream);
  if (rc) return rc;
  if ((rc = slam_set_option(h, "grad_overwrite_next", 1))) return rc; /* the first backward of a step stores its gradients */
  if (grad_image_bf16 && (rc = slam_set_grad_image(h, grad_image_bf16))) return rc;
  if ((rc = slam_backward(h, 1.0f, 4 /* decoder layers per bucket */, on_bucket, &c, stream))) return rc;
  if (c.rc) return c.rc;
  if ((rc = slam_comm_finish(h, stream))) return rc; /* `stream` waits for every exchange */
  if ((rc = slam_grad_norm(h, 0.5f, norm_out_dev, stream))) return rc;
  return slam_adamw_step_bf16(h, exp_avg_bf16, exp_avg_sq_bf16, norm_out_dev, lr, 0.9, 0.999, 1e-8, 0.0, step, 0, stream);
}

/* ---- the same step in the reduce-scatter / all-gather form (the exchange the trainer defaults to at N > 1) ------------------
 * One bucket = the whole buffer here (bucket_layers = 0: the callback fires once, with [0, n_params)); n_params is assumed to
 * be world x a multiple of the gradient-norm chunk - a real consumer cuts its buckets there and all-reduces the few thousand
 * elements above the last multiple, like slamkit_amd/trainer/dp.py does. chunk_sums: slam_param_count / slam_grad_chunk_elems
 * floats (rounded up), zeroed by the caller; all_reduce_chunk_sums: the consumer's own small all-reduce (sum) of that array. */
struct rs_ctx {
  SlamEngine* h;
  slam_stream_t stream;
  int bf16_exchange;
  int rc;
};
static void on_bucket_rs(void* user, int64_t offset, int64_t count) {
  struct rs_ctx* c = (struct rs_ctx*)user;
  slam_stream_t ready = slam_bucket_stream(c->h);
  int rc = slam_reduce_scatter_grads_async(c->h, offset, count, c->bf16_exchange, ready ? ready : c->stream);
  if (rc != SLAM_OK && c->rc == SLAM_OK) c->rc = rc;
}
int dp_step_rs_ag(SlamEngine* h, slam_stream_t stream, int rank, int world, const int64_t* ids, const int64_t* labels, int32_t B,
                  int32_t T, double global_num_items, float* loss_dev, void* grad_image_bf16, void* exp_avg_bf16,
                  void* exp_avg_sq_bf16, float* chunk_sums_dev, int (*all_reduce_chunk_sums)(float*, int64_t, slam_stream_t),
                  float* norm_out_dev, double lr, int32_t step) {
  struct rs_ctx c = {h, stream, grad_image_bf16 != NULL, SLAM_OK};
  const int64_t n = slam_param_count(h), s = n / world, mine = (int64_t)rank * s;
  const int64_t chunk = slam_grad_chunk_elems();
  int rc = slam_forward(h, ids, labels, NULL, NULL, NULL, B, T, global_num_items, loss_dev, NULL, stream);
  if (rc) return rc;
  if ((rc = slam_set_option(h, "grad_overwrite_next", 1))) return rc;
  if (grad_image_bf16 && (rc = slam_set_grad_image(h, grad_image_bf16))) return rc;
  if ((rc = slam_backward(h, 1.0f, 0, on_bucket_rs, &c, stream))) return rc;
  if (c.rc) return c.rc;
  if ((rc = slam_comm_finish(h, stream))) return rc;
  /* global norm from the chunk sums of the owned shards (disjoint support: one small all-reduce), AdamW on the owned shard */
  if ((rc = slam_grad_sumsq_chunks(h, mine, s, chunk_sums_dev, stream))) return rc;
  if ((rc = all_reduce_chunk_sums(chunk_sums_dev, (n + chunk - 1) / chunk, stream))) return rc;
  if ((rc = slam_grad_norm_from_chunks(h, chunk_sums_dev, 0.5f, norm_out_dev, stream))) return rc;
  if ((rc = slam_adamw_range_bf16(h, mine, s, (char*)exp_avg_bf16 + 2 * mine, (char*)exp_avg_sq_bf16 + 2 * mine, norm_out_dev, lr, 0.9,
                                  0.999, 1e-8, 0.0, step, 0, stream)))
    return rc;
  return slam_allgather_params_async(h, 0, n, stream); /* arrives under the next forward, which waits for it where it must */
}

/* once per process: rank 0 fills `id` with slam_comm_unique_id and broadcasts the 128 bytes by its own means */
int dp_init(SlamEngine* h, unsigned char id[SLAM_COMM_ID_BYTES], int rank, int world, int i_am_the_id_source) {
  if (i_am_the_id_source) {
    int rc = slam_comm_unique_id(id, SLAM_COMM_ID_BYTES);
    if (rc) return rc; /* SLAM_EUNSUPPORTED: no librccl.so.1 on this box */
  }
  /* ... broadcast id[] to every rank here ... */
  return slam_comm_init(h, id, rank, world);
}
