"""CPU oracle for the slamkit cli/train.py hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / reported baseline. The product path (slamkit_amd/) never imports it.

A plain fp32 PyTorch restatement (no `transformers`, no `slamkit` imports) of what the reference
executes behind `UnitLM.forward` and the HF Trainer step. Every function cites the reference
file:line it follows; paths are relative to /root/reference unless prefixed `hf:` (the container's
site-packages/transformers 5.15.0, the third-party dependency the arithmetic lives in, pinned by the
reference only as `transformers>=4.48.1`, pyproject.toml:13).

Parity status: PINNED. tests/golden/make_golden.py imports the real reference (`slamkit.model.UnitLM`
over a locally built Qwen2Config, the real UnitTokeniser / chunk_texts / HF collators) in the
authoring container and stores inputs + outputs under tests/golden/; tests/test_oracle_golden.py
checks this restatement against those vectors (fp32: logits <= 1e-5 abs, loss <= 1e-6, grads <= 1e-5
rel), and against the reference's own known-answer pair example_data/{features,tokens}.jsonl.
The optimizer step (clip_coef, cosine_with_min_lr, adamw_update / adamw_update_bf16 around forward_loss_grads) is
pinned as a LOOP since round 4: tests/golden/make_golden_traj.py runs the real slamkit.model.UnitLM under
torch.optim.AdamW + transformers.get_scheduler("cosine_with_min_lr") + clip_grad_norm_ for 200 steps in fp32 and in the
recipe's bf16 precision (traj.npz); this restatement reproduces the fp32 curve to 2e-5 over the first 60 steps and
0.15 % over all 200, and its bf16 emulation lands on the reference's bf16 curve (tests/test_oracle_golden.py).
Still "parity unpinned": the HF `Trainer` plumbing around that loop (SLAMTrainer cannot be constructed on transformers
5.x, slam_trainer.py:50 - gradient-accumulation bookkeeping and `num_items_in_batch` gathering are restated from
trainer.py) and TRL's DPO loss (TRL absent).
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, asdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# configuration
@dataclass
class OracleConfig:
    """Qwen2-shaped decoder under UnitLM (unit_lm.py:32-79; config/model/slam.yaml:4-9)."""
    n_layers: int = 24
    hidden: int = 896
    n_heads: int = 14
    n_kv_heads: int = 2
    head_dim: int = 64
    intermediate: int = 4864
    vocab: int = 502
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    pad_token_id: int = 0

    def to_dict(self):
        return asdict(self)


SLAM_358M = OracleConfig()
TINY = OracleConfig(n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64, intermediate=512)


def hf_keys(cfg: OracleConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """State-dict layout of UnitLM(Qwen2ForCausalLM), base_model_prefix 'lm' (unit_lm.py:87)."""
    H, I, hd = cfg.hidden, cfg.intermediate, cfg.head_dim
    out = [("lm.model.embed_tokens.weight", (cfg.vocab, H))]
    for l in range(cfg.n_layers):
        p = f"lm.model.layers.{l}."
        out += [
            (p + "self_attn.q_proj.weight", (cfg.n_heads * hd, H)), (p + "self_attn.q_proj.bias", (cfg.n_heads * hd,)),
            (p + "self_attn.k_proj.weight", (cfg.n_kv_heads * hd, H)), (p + "self_attn.k_proj.bias", (cfg.n_kv_heads * hd,)),
            (p + "self_attn.v_proj.weight", (cfg.n_kv_heads * hd, H)), (p + "self_attn.v_proj.bias", (cfg.n_kv_heads * hd,)),
            (p + "self_attn.o_proj.weight", (H, cfg.n_heads * hd)),
            (p + "mlp.gate_proj.weight", (I, H)), (p + "mlp.up_proj.weight", (I, H)), (p + "mlp.down_proj.weight", (H, I)),
            (p + "input_layernorm.weight", (H,)), (p + "post_attention_layernorm.weight", (H,)),
        ]
    out.append(("lm.model.norm.weight", (H,)))
    return out


def _hash_uniform(n: int, seed: int) -> np.ndarray:
    """Deterministic uniform(-1,1) stream from a splitmix64 counter hash (no RNG library state)."""
    with np.errstate(over="ignore"):
        s0 = np.array([seed], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        x = np.arange(n, dtype=np.uint64) + s0[0]
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return (2.0 * u - 1.0)


def _hash_uniform_t(n: int, seed: int, scale: float, offset: float, dtype) -> torch.Tensor:
    """offset + scale * _hash_uniform(n, seed) cast to `dtype`, bit for bit, computed with torch (threaded, in place, in
    chunks): the numpy form above makes ten full-size 64-bit temporaries on one core - 91 s for the 358M parameters of
    Slam-358M on an 8-core container, most of the wall time of the tests that build a full-size model. int64 arithmetic
    wraps like uint64; the logical right shifts are arithmetic shifts with the sign extension masked off."""
    M64 = (1 << 64) - 1

    def i64(v):  # python int (mod 2^64) -> the int64 with the same bits
        v &= M64
        return v - (1 << 64) if v >= (1 << 63) else v

    c_add = i64(seed * 0x9E3779B97F4A7C15 + 0x9E3779B97F4A7C15)
    out = torch.empty(n, dtype=dtype)
    step = 1 << 20
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        x = torch.arange(lo, hi, dtype=torch.int64)
        x += c_add
        for sh, mul in ((30, 0xBF58476D1CE4E5B9), (27, 0x94D049BB133111EB)):
            y = (x >> sh) & ((1 << (64 - sh)) - 1)
            x ^= y
            x *= i64(mul)
        y = (x >> 31) & ((1 << 33) - 1)
        x ^= y
        x = (x >> 11) & ((1 << 53) - 1)
        u = x.to(torch.float64)
        u /= float(1 << 53)
        u *= 2.0
        u -= 1.0            # the uniform(-1, 1) value of _hash_uniform
        if scale != 1.0:
            u *= scale
        if offset != 0.0:
            u += offset
        out[lo:hi] = u.to(dtype)
    return out


def init_weights(cfg: OracleConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.0,
                 norm_jitter: float = 0.0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic weights (N(0,0.02)-scale matrices, biases 0, norms 1,
    E[pad]=0). Uniform with matching std, from a counter hash so fixtures need not store them."""
    sd = {}
    for i, (k, shp) in enumerate(hf_keys(cfg)):
        n = int(np.prod(shp))
        if k.endswith("layernorm.weight") or k.endswith("norm.weight"):
            t = _hash_uniform_t(n, seed * 1000 + i, norm_jitter, 1.0, dtype) if norm_jitter else torch.ones(n, dtype=dtype)
        elif k.endswith(".bias"):
            t = _hash_uniform_t(n, seed * 1000 + i, bias_std * math.sqrt(3.0), 0.0, dtype) if bias_std else torch.zeros(n, dtype=dtype)
        else:
            t = _hash_uniform_t(n, seed * 1000 + i, std * math.sqrt(3.0), 0.0, dtype)
        t = t.reshape(shp)
        if k == "lm.model.embed_tokens.weight" and cfg.pad_token_id is not None and cfg.pad_token_id >= 0:
            t[cfg.pad_token_id].zero_()  # nn.Embedding(padding_idx) zero row, hf: modeling_qwen2.py:327
        sd[k] = t
    return sd


# --------------------------------------------------------------------------------------------
# model forward (hf: transformers/models/qwen2/modeling_qwen2.py)
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2RMSNorm.forward, hf: modeling_qwen2.py:247-252."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float, dtype=torch.float32):
    """Qwen2RotaryEmbedding.forward, hf: modeling_qwen2.py:91-102 (default rope, scaling 1)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float32) / head_dim))
    freqs = position_ids[:, :, None].to(torch.float32) * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    """hf: modeling_qwen2.py:105-109."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """apply_rotary_pos_emb, hf: modeling_qwen2.py:112-135 (unsqueeze_dim=1)."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def segment_starts(position_ids: torch.Tensor) -> torch.Tensor:
    """Index of the first token of each token's packed sequence (position_ids == 0 restarts),
    the information flash-attn varlen derives cu_seqlens from for DataCollatorWithFlattening
    batches (hf_dataset.py:61-62; train.py:43-45)."""
    B, T = position_ids.shape
    idx = torch.arange(T).expand(B, T)
    start = torch.where(position_ids == 0, idx, torch.zeros_like(idx))
    return torch.cummax(start, dim=1).values


def attention_mask_bool(B, T, attention_mask=None, position_ids=None, packed=False):
    """Causal (+ key padding | segment) mask, eager semantics hf: modeling_qwen2.py:150-172,377-379."""
    i = torch.arange(T)
    m = (i[None, :] <= i[:, None])[None].expand(B, T, T).clone()
    if attention_mask is not None:
        m &= attention_mask.bool()[:, None, :]
    if packed and position_ids is not None:
        seg = segment_starts(position_ids)
        m &= (i[None, None, :] >= seg[:, :, None])
    return m


_FUSED_ATTENTION = True  # False: always the eager expression (equivalence test)


def attention(q, k, v, mask, scale, bf16_probs: bool = False):
    """eager_attention_forward with repeat_kv, hf: modeling_qwen2.py:138-172 (softmax in fp32; the probabilities are cast
    to the value dtype before P V - emulated with bf16_probs when the tensors are fp32 stand-ins for bf16 ones)."""
    B, nH, T, hd = q.shape
    if not bf16_probs and q.dtype == torch.float32 and _FUSED_ATTENTION:
        # The same function through torch's fused CPU kernel (the call HF's default attn_implementation="sdpa" makes:
        # modeling_qwen2.py -> sdpa_attention_forward): additive mask with finfo.min like the eager path, fp32 softmax, GQA by
        # head broadcast. No [B, nH, T, T] tensors: 2-3x less wall time for the full-size models of the GPU parity tests; held to
        # the eager expression above within 1e-6 (tests/test_oracle_golden.py) and to the reference goldens like before.
        amask = torch.zeros(mask.shape[0], 1, T, T, dtype=q.dtype).masked_fill_(~mask[:, None], torch.finfo(q.dtype).min)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=amask, scale=scale, enable_gqa=k.shape[1] != nH)
        return o.transpose(1, 2).contiguous()
    rep = nH // k.shape[1]
    k = k.repeat_interleave(rep, dim=1)
    v = v.repeat_interleave(rep, dim=1)
    s = torch.matmul(q, k.transpose(2, 3)) * scale
    s = s.masked_fill(~mask[:, None], torch.finfo(s.dtype).min)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    if bf16_probs:
        p = _bf16_round(p)
    return torch.matmul(p, v).transpose(1, 2).contiguous()


def _bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Value rounded to bf16 (kept in fp32 storage), identity gradient."""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def decoder_stack(cfg: OracleConfig, sd: Dict[str, torch.Tensor], h: torch.Tensor, E_head: torch.Tensor,
                  attention_mask=None, position_ids=None, packed=False, bf16_acts: bool = False, collect: Optional[list] = None):
    """Qwen2Model layers + final norm + tied head on given input embeddings
    (hf: modeling_qwen2.py:342-402, DecoderLayer :269-298, Attention :189-233, MLP :41-48, head :465).
    bf16_acts=True restates the reference's OWN precision (bf16 parameters under bf16 autocast, slam.yaml:9 +
    training_args bf16): every module output the HF path materialises as a bf16 tensor - Linear outputs, RMSNorm
    outputs, rotated q / k, attention probabilities and output, the SwiGLU product, residual sums, logits - is rounded
    to bf16 at that point (arithmetic inside a module stays fp32, as the GPU kernels accumulate). Used to calibrate how
    far ANY bf16 implementation sits from the fp32 run at a given depth / width (tests/test_gpu_model.py).
    collect: list that receives HF's `output_hidden_states` tuple - the input of every decoder layer, then the output of
    the final norm (tests/golden/make_golden_deep.py pins the emulation against the reference per depth)."""
    r = _bf16_round if bf16_acts else (lambda t: t)
    B, T, _ = h.shape
    hd, nH, nKV = cfg.head_dim, cfg.n_heads, cfg.n_kv_heads
    if position_ids is None:
        position_ids = torch.arange(T)[None].expand(B, T)
    cos, sin = rope_cos_sin(position_ids, hd, cfg.rope_theta, h.dtype)
    mask = attention_mask_bool(B, T, attention_mask, position_ids, packed)
    for l in range(cfg.n_layers):
        p = f"lm.model.layers.{l}."
        if collect is not None:
            collect.append(h.detach())
        x = r(rms_norm(h, sd[p + "input_layernorm.weight"], cfg.rms_eps))
        q = r(F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])).view(B, T, nH, hd).transpose(1, 2)
        k = r(F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])).view(B, T, nKV, hd).transpose(1, 2)
        v = r(F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])).view(B, T, nKV, hd).transpose(1, 2)
        q, k = apply_rope(q, k, cos, sin)
        q, k = r(q), r(k)
        a = r(attention(q, k, v, mask, hd ** -0.5, bf16_probs=bf16_acts).reshape(B, T, nH * hd))
        h = r(h + r(F.linear(a, sd[p + "self_attn.o_proj.weight"])))
        x = r(rms_norm(h, sd[p + "post_attention_layernorm.weight"], cfg.rms_eps))
        g = r(F.linear(x, sd[p + "mlp.gate_proj.weight"]))
        u = r(F.linear(x, sd[p + "mlp.up_proj.weight"]))
        h = r(h + r(F.linear(r(F.silu(g) * u), sd[p + "mlp.down_proj.weight"])))
    hf = r(rms_norm(h, sd["lm.model.norm.weight"], cfg.rms_eps))
    if collect is not None:
        collect.append(hf.detach())
    return r(F.linear(hf, E_head))  # tied lm_head, hf: modeling_qwen2.py:407,465


def model_forward(cfg: OracleConfig, sd: Dict[str, torch.Tensor], input_ids: torch.Tensor,
                  attention_mask: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                  packed: bool = False, bf16_acts: bool = False, collect: Optional[list] = None):
    """UnitLM.forward -> Qwen2ForCausalLM.forward without labels (unit_lm.py:155-167)."""
    E = sd["lm.model.embed_tokens.weight"]
    return decoder_stack(cfg, sd, F.embedding(input_ids, E), E, attention_mask, position_ids, packed, bf16_acts, collect)


def compute_loss(logits: torch.Tensor, labels: torch.Tensor, num_items_in_batch=None, ignore_index: int = -100):
    """slamkit/model/unit_lm.py:13-29 restated line by line."""
    logits = logits.float()
    shift_logits = logits[..., :-1, :].contiguous().view(-1, logits.size(-1))
    shift_labels = labels[..., 1:].contiguous().view(-1)
    reduction = "sum" if num_items_in_batch is not None else "mean"
    loss = F.cross_entropy(shift_logits, shift_labels, reduction=reduction, ignore_index=ignore_index)
    if reduction == "sum":
        loss = loss / num_items_in_batch
    return loss


def calc_nll(logits, target, mask, len_norm=True):
    """slamkit/utils/calculation_utils.py:5-29."""
    losses = F.cross_entropy(logits.contiguous().view(-1, logits.size(-1)), target.long().contiguous().view(-1),
                             reduction="none").view(*target.size())
    ll = (losses * mask).sum(dim=-1)
    return ll / mask.sum(dim=-1) if len_norm else ll


def log_likelihood(cfg, sd, tokens: torch.Tensor, mean_nll: bool, ignore_tokens=None):
    """UnitLM.log_likelihood, unit_lm.py:184-194."""
    with torch.no_grad():
        logits = model_forward(cfg, sd, tokens)
        if ignore_tokens is not None:
            logits[:, :, ignore_tokens] = float("-inf")
        shifted_x = tokens[..., 1:].clone()
        shifted_logits = logits[..., :-1, :]
        shifted_x[shifted_x == cfg.pad_token_id] = -100
        mask = shifted_x.ne(-100)
        tgt = shifted_x.clamp(min=0)
        return -calc_nll(shifted_logits, tgt, mask, mean_nll)


def forward_loss_grads(cfg, sd, input_ids, labels, attention_mask=None, position_ids=None, packed=False,
                       num_items_in_batch=None, bf16_acts: bool = False):
    """One fwd + loss + autograd backward; grads keyed like `sd`. F.embedding(padding_idx=pad)
    suppresses the gather-side gradient of the pad row exactly like nn.Embedding(padding_idx)
    (hf: modeling_qwen2.py:327); the tied head still contributes to that row (SURVEY.md §7 iii)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    E = params["lm.model.embed_tokens.weight"]
    pad = cfg.pad_token_id if (cfg.pad_token_id is not None and cfg.pad_token_id >= 0) else None
    h0 = F.embedding(input_ids, E, padding_idx=pad)
    logits = decoder_stack(cfg, params, h0, E, attention_mask, position_ids, packed, bf16_acts)
    loss = compute_loss(logits, labels, num_items_in_batch)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad.detach() for k, v in params.items()}


# --------------------------------------------------------------------------------------------
# tokeniser / data (integer work; bit-exact bar)
def unit_vocab(num_units=500, pad_id=0, bos_eos_id=1) -> Dict[str, int]:
    """UnitTokeniser._init_text_tokeniser vocab, unit_tokeniser.py:33-38."""
    offset = max(bos_eos_id, pad_id) + 1
    v = {f"<Un{i}>": i + offset for i in range(num_units)}
    v.update({"<PAD>": pad_id, "<S>": bos_eos_id})
    return v


def stringify_units(units: Sequence[int]) -> str:
    """UnitTokeniser.stringify_representation, unit_tokeniser.py:62-63."""
    return "".join(f"<Un{u}>" for u in units)


def unit_tokenise(audio_repr: str, num_units=500, pad_id=0, bos_eos_id=1) -> Dict[str, List[int]]:
    """prepare_sample -> text_tokeniser(audio_repr): split on '>' (merged_with_previous), WordLevel
    lookup, '<S> $0 <S>' template (unit_tokeniser.py:39-47, 82-83)."""
    vocab = unit_vocab(num_units, pad_id, bos_eos_id)
    toks = [t for t in re.findall(r"[^>]*>", audio_repr)]
    ids = [bos_eos_id] + [vocab[t] for t in toks] + [bos_eos_id]
    return {"input_ids": ids, "attention_mask": [1] * len(ids)}


def split_into_chunks(lst, chunk_size):
    """hf_dataset.py:16-18."""
    return [lst[i:i + chunk_size] for i in range(0, len(lst), chunk_size)]


def chunk_texts(examples: Dict[str, List[List[int]]], chunk_size: int):
    """hf_dataset.py:21-26 (remainders kept, no special tokens re-added)."""
    return {k: [c for l in v for c in split_into_chunks(l, chunk_size)] for k, v in examples.items()}


def collate_lm(features: List[Dict[str, List[int]]], pad_id=0):
    """DataCollatorForLanguageModeling(mlm=False): right-pad, labels = ids with pad -> -100
    (hf_dataset.py:64)."""
    T = max(len(f["input_ids"]) for f in features)
    ids = torch.full((len(features), T), pad_id, dtype=torch.long)
    am = torch.zeros((len(features), T), dtype=torch.long)
    for i, f in enumerate(features):
        n = len(f["input_ids"])
        ids[i, :n] = torch.tensor(f["input_ids"])
        am[i, :n] = 1
    labels = ids.clone()
    labels[labels == pad_id] = -100
    return {"input_ids": ids, "attention_mask": am, "labels": labels}


def collate_flatten(features: List[Dict[str, List[int]]]):
    """DataCollatorWithFlattening: one [1, sum T] row, position_ids restart per sequence,
    labels[first token of each sequence] = -100 (hf_dataset.py:61-62)."""
    ids, pos, lab = [], [], []
    for f in features:
        x = list(f["input_ids"])
        ids += x
        pos += list(range(len(x)))
        lab += [-100] + x[1:]
    return {"input_ids": torch.tensor([ids]), "position_ids": torch.tensor([pos]), "labels": torch.tensor([lab])}


def get_num_tokens(labels: torch.Tensor, min_id=None, max_id=None) -> int:
    """SLAMTrainer.get_num_tokens, slam_trainer.py:59-65."""
    v = labels != -100
    if min_id is not None:
        v &= labels >= min_id
    if max_id is not None:
        v &= labels <= max_id
    return int(v.sum())


def dpo_tokenize_row(prompt_ids, chosen_ids, rejected_ids, bos=1, eos=1, max_prompt_length=None,
                     max_completion_length=None):
    """SLAMDPOTrainer.tokenize_row tail, slam_dpo_trainer.py:39-64 (ids already tokenised w/o specials)."""
    p = [bos] + list(prompt_ids)
    c = list(chosen_ids) + [eos]
    r = list(rejected_ids) + [eos]
    if max_prompt_length is not None:
        p = p[-max_prompt_length:]
    if max_completion_length is not None:
        c, r = c[:max_completion_length], r[:max_completion_length]
    return {"prompt_input_ids": p, "chosen_input_ids": c, "rejected_input_ids": r}


# --------------------------------------------------------------------------------------------
# optimiser step (hf Trainer semantics) - restated from definitions
def cosine_with_min_lr(step: int, warmup: int, total: int, min_lr_rate: float, num_cycles: float = 0.5) -> float:
    """hf: optimization.py `_get_cosine_with_min_lr_schedule_with_warmup_lr_lambda`."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    prog = float(step - warmup) / float(max(1, total - warmup))
    f = 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog))
    f = f * (1 - min_lr_rate) + min_lr_rate
    return max(0.0, f)


def clip_coef(grads: Dict[str, torch.Tensor], max_norm: float) -> Tuple[float, float]:
    """torch.nn.utils.clip_grad_norm_: total L2 norm, coef = min(1, max_norm/(norm+1e-6))."""
    tot = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))
    return tot, min(1.0, max_norm / (tot + 1e-6))


def adamw_update(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.AdamW single-tensor update (fp32), in place."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def adamw_update_bf16_moments(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """The middle optimizer precision of the engine (slam_adamw_step_bf16_moments; no counterpart in the reference: its
    recipe keeps everything in bf16, torch's default everything in fp32): fp32 master `p`, Adam moments STORED in bf16 -
    widened to fp32, updated with adamw_update's own expressions, rounded once when stored; the parameter step uses
    the unrounded fp32 moments of this step."""
    assert p.dtype == torch.float32 and m.dtype == v.dtype == torch.bfloat16
    f = torch.float32
    mf, vf, gf = m.to(f), v.to(f), g.to(f)
    p.mul_(1 - lr * wd)
    mf = mf * b1 + gf * (1 - b1)
    vf = vf * b2 + gf * gf * (1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    den = vf.sqrt() / math.sqrt(bc2) + eps
    p.addcdiv_(mf, den, value=-(lr / bc1))
    m.copy_(mf.to(torch.bfloat16))
    v.copy_(vf.to(torch.bfloat16))


def adamw_update_bf16(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.AdamW(fused=True) on bf16 parameters with bf16 moments, in place - the optimizer precision of the
    Slam recipe (/root/reference config/model/slam.yaml:9 `torch_dtype: bfloat16`; the HF Trainer builds AdamW on the
    bf16 parameters, so exp_avg / exp_avg_sq are bf16 too). Restated from the fused kernel's semantics: every element
    is widened to fp32, exp_avg moves by lerp(exp_avg, g, 1 - b1), exp_avg_sq = b2 v + (1 - b2) g^2, the parameter step
    uses lr / (1 - b1^t) and sqrt(v) / sqrt(1 - b2^t) + eps, and each tensor is rounded to bf16 once when stored.
    `g` may be fp32 (the engine keeps fp32 gradients) or bf16. Pinned against torch's own fused CPU kernel by
    tests/test_oracle_golden.py::test_adamw_bf16_state_matches_torch_fused."""
    assert p.dtype == m.dtype == v.dtype == torch.bfloat16
    f = torch.float32
    pf, mf, vf, gf = p.to(f), m.to(f), v.to(f), g.to(f)
    pf = pf * torch.tensor(1 - lr * wd, dtype=f)
    mf = mf + torch.tensor(1 - b1, dtype=f) * (gf - mf)
    vf = torch.tensor(b2, dtype=f) * vf + torch.tensor(1 - b2, dtype=f) * gf * gf
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    den = vf.sqrt() / torch.tensor(math.sqrt(bc2), dtype=f) + torch.tensor(eps, dtype=f)
    pf = pf - torch.tensor(lr / bc1, dtype=f) * (mf / den)
    p.copy_(pf.to(torch.bfloat16))
    m.copy_(mf.to(torch.bfloat16))
    v.copy_(vf.to(torch.bfloat16))


def dpo_loss(pi_c, pi_r, ref_c, ref_r, beta=0.1):
    """Sigmoid DPO loss from its definition (TRL absent here - parity unpinned)."""
    return -F.logsigmoid(beta * ((pi_c - pi_r) - (ref_c - ref_r))).mean()
