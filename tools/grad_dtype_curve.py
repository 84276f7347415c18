"""Round 6 evidence run (tools only): does keeping the step's FINAL gradients in bf16 (the reference's own gradient precision) change
training at the Slam-358M scale? 300 optimizer steps of Slam-358M (B 8 x T 1024, bf16 optimizer state, clip 0.5, cosine_with_min_lr,
warm-up 30) on a learnable synthetic stream - rows ids[t+1] = ids[t] + stride (mod 500) with per-row start / stride and 10 % of the
positions replaced by uniform noise - three times from the same initial weights and the same batches:
  bf16   : grad_dtype bfloat16 (default with bf16 state): final gradients stored in bf16 only, norm from backward's partial sums
  fp32   : grad_dtype float32: fp32 final gradients, norm from the partial sums
  round5 : fp32 gradients + the chunked norm pass (the round-5 step)
Prints a markdown table of the loss every 20 steps and the max / mean relative deviation of the bf16 curve from the other two.
Usage: python tools/grad_dtype_curve.py [steps] > profiles/r6_grad_dtype_curve.md"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd.model import UnitLM, UnitLMConfig
from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
from slamkit_amd.trainer.training_args import lr_lambda

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, T, V = 8, 1024, 502
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def batch(i):
    g = torch.Generator().manual_seed(1000 + i)
    start = torch.randint(0, 500, (B, 1), generator=g)
    stride = torch.tensor([1, 3, 7, 11])[torch.randint(0, 4, (B, 1), generator=g)]
    ids = (start + stride * torch.arange(T)[None]) % 500 + 2
    noise = torch.rand(B, T, generator=g) < 0.10
    ids = torch.where(noise, torch.randint(2, V, (B, T), generator=g), ids)
    ids[:, 0] = 1
    return {"input_ids": ids.to(dev), "labels": ids.to(dev)}


def run(grad_dtype, final_mode=None):
    model = UnitLM(UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", rope_theta=10000.0, vocab_size=V, max_tokens=B * T), seed=0)
    args = SLAMTrainingArguments(per_device_train_batch_size=B, learning_rate=1e-3, max_grad_norm=0.5, logging_steps=0, warmup_steps=30,
                                 optim_state_dtype="bfloat16", grad_dtype=grad_dtype)
    tr = SLAMTrainer(model=model, args=args)
    if final_mode is not None:
        tr._final_mode = final_mode
    losses, norms = [], []
    n = float(B * T)
    for s in range(STEPS):
        tr._loss_acc.zero_()
        tr.optimizer_step([batch(s)], args.learning_rate * lr_lambda(args, s, STEPS), counts=(n, n))
        losses.append(float(tr._loss_acc))
        norms.append(float(tr.norm_out[0]))
    del tr, model
    torch.cuda.empty_cache()
    return losses, norms


res = {"bf16": run("bfloat16"), "fp32": run("float32"), "round5": run("float32", 0)}
print("# Round 6: final gradients in bf16 vs fp32 - Slam-358M, 300 steps on a learnable stream (tools/grad_dtype_curve.py)\n")
print(__doc__.split("Prints")[0].strip().replace("\n", " ") + "\n")
print("| step | loss, bf16 final gradients | loss, fp32 final gradients | loss, round-5 step | pre-clip grad norm (bf16 / fp32 / round 5) |")
print("|---|---|---|---|---|")
for s in [0, 1, 2, 5, 10] + list(range(20, STEPS, 20)) + [STEPS - 1]:
    print(f"| {s + 1} | {res['bf16'][0][s]:.4f} | {res['fp32'][0][s]:.4f} | {res['round5'][0][s]:.4f} | "
          f"{res['bf16'][1][s]:.4f} / {res['fp32'][1][s]:.4f} / {res['round5'][1][s]:.4f} |")
a = torch.tensor(res["bf16"][0])
for k in ("fp32", "round5"):
    b = torch.tensor(res[k][0])
    rel = ((a - b).abs() / b)
    print(f"\nbf16 vs {k}: max relative loss deviation {float(rel.max()):.3%} (step {int(rel.argmax()) + 1}), mean {float(rel.mean()):.3%}, "
          f"last 50 steps mean {float(rel[-50:].mean()):.3%}; final loss {a[-1]:.4f} vs {b[-1]:.4f}")
b, c = torch.tensor(res["fp32"][0]), torch.tensor(res["round5"][0])
rel = (b - c).abs() / c
print(f"\nfp32 vs round5 (same gradients, norm summed in another order): max {float(rel.max()):.3%}, mean {float(rel.mean()):.3%}")
