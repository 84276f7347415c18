"""-m gpu: the last backward of an optimizer step keeps its final gradient values the way the reference does - in bf16
(bf16 parameters have bf16 .grad: /root/reference config/model/slam.yaml:9) - and emits the gradient-norm partials from the
same stores (engine option "grad_final_next", include/slam_engine.h). Bit-exact contracts against the plain fp32 backward of
the same library: the bf16 values are the round-to-nearest-even of the fp32 ones, tensor by tensor, whichever kernel stores
the final value (unsplit 256 x 224 tiles in both orientations, 128 x 128 tiles with and without slab reduces, norm / bias
finish, one-hot embedding GEMM, large-vocabulary scatter + conversion pass); the norm is the norm of what was stored."""
import pytest
import torch

from oracle import slam_oracle as O
from tests.test_gpu_model import _mk

pytestmark = pytest.mark.gpu

# Slam-358M's matrix shapes (config/model/slam.yaml:4-9 -> Qwen2.5-0.5B dims) at two layers: every weight-gradient kernel
# of the headline step takes the plan it takes there (M = 8 x 1024 tokens)
SLAM2 = O.OracleConfig(vocab=502, hidden=896, n_layers=2, n_heads=14, n_kv_heads=2, head_dim=64, intermediate=4864)
BIGV = O.OracleConfig(vocab=5003, hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, intermediate=512)
# two layers of the configs[3]-shaped body (Qwen2.5-1.5B dims, head_dim 128): the 256 x 224 weight-gradient kernel in its other tile
# counts and orientations (Wgu 17920 x 1536 transposed-store, Wd 1536 x 8960 direct), 128 x 128 plans of other sizes
QW2 = O.OracleConfig(vocab=700, hidden=1536, n_layers=2, n_heads=12, n_kv_heads=2, head_dim=128, intermediate=8960, rope_theta=1e6)


def _batch(cfg, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(2, cfg.vocab, (B, T), generator=g)
    ids[:, 0] = 1
    return ids


def _run(m, batches, final):
    """GA over `batches`: the first backward overwrites, the last one is `final`; returns (fp32 buffer, bf16 buffer or None,
    [norm, clip])."""
    eng = m.engine
    n_items = float(sum(b.numel() for b in batches))
    for i, ids in enumerate(batches):
        if i == 0:
            eng.set_option("grad_overwrite_next", 1)
        m(input_ids=ids, labels=ids, num_items_in_batch=n_items, return_logits=False)
        m.backward(1.0, final=final if i == len(batches) - 1 else 0)
    norm = torch.zeros(2, device="cuda")
    eng.grad_norm(0.5, norm)
    torch.cuda.synchronize()
    return m.flat_grads.clone(), (m.flat_grads16.clone() if final == 2 else None), norm.cpu()


@pytest.mark.parametrize("cfg,B,T", [(O.TINY, 2, 96), (SLAM2, 8, 1024), (SLAM2, 3, 704), (BIGV, 2, 320), (QW2, 2, 2048)],
                         ids=["tiny", "slam2", "slam2_ragged", "bigvocab", "qwen1p5b_2layers"])
@pytest.mark.parametrize("ga", [1, 2])
def test_final_bf16_gradients_are_the_rounded_fp32_ones(cfg, B, T, ga):
    sd = O.init_weights(cfg, seed=5, bias_std=0.02, norm_jitter=0.05)
    m = _mk(cfg, sd, max_tokens=B * T)
    batches = [_batch(cfg, B, T, 10 + i) for i in range(ga)]
    g32, _, n0 = _run(m, batches, final=0)
    g32b, none, n1 = _run(m, batches, final=1)
    assert torch.equal(g32, g32b), "final = 1 changed the fp32 gradients"
    m.flat_grads.fill_(float("nan")) if ga == 1 else None  # GA 1: nothing may read the fp32 buffer but the embedding's head half
    stale, g16, n2 = _run(m, batches, final=2)
    want = g32.to(torch.bfloat16)
    assert torch.equal(g16, want), f"{int((g16 != want).sum())} of {want.numel()} bf16 gradients differ from RNE(fp32)"
    # norms: the chunked fp32 pass, the partials of the fp32 stores, the partials of the bf16 stores
    ref32 = float(g32.double().norm())
    ref16 = float(want.double().norm())
    print(f"[parity] {cfg.hidden}x{cfg.n_layers} ga{ga}: norm chunks {float(n0[0]):.7e} partials fp32 {float(n1[0]):.7e} (fp64 {ref32:.7e}) "
          f"partials bf16 {float(n2[0]):.7e} (fp64 {ref16:.7e})")
    assert abs(float(n0[0]) - ref32) <= 2e-6 * ref32
    assert abs(float(n1[0]) - ref32) <= 2e-6 * ref32
    assert abs(float(n2[0]) - ref16) <= 2e-6 * ref16
    for nn, ref in ((n1, ref32), (n2, ref16)):
        assert abs(float(nn[1]) - min(1.0, 0.5 / (ref + 1e-6))) <= 1e-5
    # the same bits every run
    _, g16b, n2b = _run(m, batches, final=2)
    assert torch.equal(g16, g16b) and torch.equal(n2, n2b)
    # named_grads follows the buffer the last backward wrote
    k = "lm.model.layers.1.mlp.down_proj.weight"
    assert torch.equal(dict(m.named_grads())[k], m._view(want, k).float())


@pytest.mark.parametrize("osd", ["bfloat16", "float32", "float32_bf16_moments"])
def test_adamw_from_bf16_gradients_equals_adamw_from_their_fp32_copy(osd):
    """The optimizer kernels read bf16 gradients through the same arithmetic: an update from the bf16 buffer must leave the
    bits an update from the widened copy of those values leaves (all three state precisions, matrices and vectors)."""
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
    cfg = O.TINY
    sd = O.init_weights(cfg, seed=4, bias_std=0.02, norm_jitter=0.05)
    ids = [_batch(cfg, 2, 96, 3), _batch(cfg, 2, 96, 4)]
    res = []
    for mode in ("bf16_buffer", "widened"):
        m = _mk(cfg, sd, max_tokens=192)
        tr = SLAMTrainer(model=m, args=SLAMTrainingArguments(optim_state_dtype=osd, grad_dtype="bfloat16", weight_decay=0.01,
                                                             max_grad_norm=0.5, logging_steps=0))
        assert tr._final_mode == 2
        for step in range(3):
            if mode == "bf16_buffer":
                tr.optimizer_step([{"input_ids": ids[step % 2], "labels": ids[step % 2]}], 1e-3)
            else:  # the same step by hand: final bf16 gradients, widened into the fp32 buffer, plain clip + AdamW over it
                m.engine.set_option("grad_overwrite_next", 1)
                m(input_ids=ids[step % 2], labels=ids[step % 2], num_items_in_batch=float(ids[0].numel()), return_logits=False)
                m.backward(1.0, final=2)
                norm_p = torch.zeros(2, device="cuda")
                m.engine.grad_norm(0.5, norm_p)  # from the partials, like the trainer's
                m.flat_grads.copy_(m.flat_grads16.float())
                m.engine.set_option("grad_overwrite_next", 1)  # a plain backward resets the engine to the fp32 buffer ...
                m(input_ids=ids[step % 2], labels=ids[step % 2], num_items_in_batch=float(ids[0].numel()), return_logits=False)
                m.backward(1.0)
                m.flat_grads.copy_(m.flat_grads16.float())          # ... which holds the widened bf16 values
                tr.norm_out.copy_(norm_p)
                tr.opt_step += 1
                a = tr.args
                if tr.state_dtype == torch.bfloat16:
                    m.engine.adamw_step_bf16(tr.exp_avg, tr.exp_avg_sq, tr.norm_out, 1e-3, a.adam_beta1, a.adam_beta2, a.adam_epsilon,
                                             a.weight_decay, tr.opt_step, zero_grad=False)
                else:
                    m.engine.adamw_step(m.flat_master, tr.exp_avg, tr.exp_avg_sq, tr.norm_out, 1e-3, a.adam_beta1, a.adam_beta2,
                                        a.adam_epsilon, a.weight_decay, tr.opt_step, zero_grad=False)
        torch.cuda.synchronize()
        res.append((m.flat_params.clone(), m._weights.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), m.flat_params_t.clone()))
    for a, b, name in zip(res[0], res[1], ("params", "weights", "exp_avg", "exp_avg_sq", "params_t")):
        assert torch.equal(a, b), name


def test_trainer_grad_dtype_default_and_fp32_norm_partials():
    """grad_dtype follows the optimizer state by default (bf16 state -> bf16 final gradients); with fp32 final gradients the
    step differs from the round-5 step (chunked norm pass) only through the summation order of the norm."""
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
    cfg = O.TINY
    sd = O.init_weights(cfg, seed=4, bias_std=0.02, norm_jitter=0.05)
    ids = _batch(cfg, 2, 96, 3)
    m = _mk(cfg, sd, max_tokens=192)
    assert SLAMTrainer(model=m, args=SLAMTrainingArguments(optim_state_dtype="bfloat16", logging_steps=0))._final_mode == 2
    m = _mk(cfg, sd, max_tokens=192)
    tr = SLAMTrainer(model=m, args=SLAMTrainingArguments(optim_state_dtype="float32", logging_steps=0, max_grad_norm=0.5))
    assert tr._final_mode == 1
    tr.optimizer_step([{"input_ids": ids, "labels": ids}], 1e-3)
    torch.cuda.synchronize()
    n_partials = float(tr.norm_out[0])
    ref = float(m.flat_grads.double().norm())
    assert abs(n_partials - ref) <= 2e-6 * ref
    with pytest.raises(Exception):
        m.engine.set_option("grad_final_next", 3)
    m(input_ids=ids, labels=ids, return_logits=False)
    m.engine.set_option("grad_final_next", 2)
    with pytest.raises(Exception, match="slam_set_grad_image"):
        m.engine.backward(1.0)
