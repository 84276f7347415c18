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

/* once per process: rank 0 fills `id` with slam_comm_unique_id and broadcasts the 128 bytes by its own means */
int dp_init(SlamEngine* h, unsigned char id[SLAM_COMM_ID_BYTES], int rank, int world, int i_am_the_id_source) {
  if (i_am_the_id_source) {
    int rc = slam_comm_unique_id(id, SLAM_COMM_ID_BYTES);
    if (rc) return rc; /* SLAM_EUNSUPPORTED: no librccl.so.1 on this box */
  }
  /* ... broadcast id[] to every rank here ... */
  return slam_comm_init(h, id, rank, world);
}
