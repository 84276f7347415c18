"""Engine-backed UnitLM: the plugin surface of /root/reference slamkit/model/unit_lm.py
(UnitLMConfig :32-79, UnitLM :82-212, compute_loss :13-29) with the HuggingFace Qwen2 forward /
autograd backward replaced by the gfx950 HIP engine (include/slam_engine.h).

Same call shapes: `model(input_ids, attention_mask, position_ids, labels, num_items_in_batch=...)`
returns an object with `.loss` (fp32 scalar, `.backward()` works) and `.logits [B,T,V]` (bf16);
`log_likelihood`, `generate`, `save_pretrained / from_pretrained` (HF state-dict key layout
`lm.model.layers.{i}.self_attn.q_proj.weight` ..., SURVEY.md §5 checkpoint row) and
`get_input_embeddings / get_output_embeddings` exist. There is no eager fallback: without the
HIP library or a GPU, construction fails.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import torch

from .. import engine as E
from .token_lm import TokenLM

# Qwen2.5-0.5B body (config/model/slam.yaml:7 base_model_name) - the hub is unreachable from a
# training box, so the dims the reference pulls via AutoConfig.from_pretrained (unit_lm.py:64)
# are carried here.
KNOWN_BASE_CONFIGS = {
    "Qwen/Qwen2.5-0.5B": dict(num_hidden_layers=24, hidden_size=896, num_attention_heads=14, num_key_value_heads=2,
                              head_dim=64, intermediate_size=4864, rms_norm_eps=1e-6, rope_theta=1000000.0,
                              tie_word_embeddings=True, initializer_range=0.02),
    # interleaved speech-text scale-up body (BASELINE.json configs[3]; dims restated in SURVEY.md §8a-note)
    "Qwen/Qwen2.5-1.5B": dict(num_hidden_layers=28, hidden_size=1536, num_attention_heads=12, num_key_value_heads=2,
                              head_dim=128, intermediate_size=8960, rms_norm_eps=1e-6, rope_theta=1000000.0,
                              tie_word_embeddings=True, initializer_range=0.02),
}


_HF_DIM_KEYS = ("num_hidden_layers", "hidden_size", "num_attention_heads", "num_key_value_heads", "head_dim",
                "intermediate_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings", "initializer_range")


def base_config_from_hf(c: dict) -> dict:
    """Engine-side `base_config` from a HuggingFace Qwen2 config dict (a text-LM `config.json`, or the `base_config`
    object the reference's UnitLMConfig serialises, unit_lm.py:63-73). transformers 4.x stores `rope_theta` at the top
    level, 5.x under `rope_parameters`."""
    mt = c.get("model_type", "qwen2")
    if mt not in ("qwen2",):
        raise ValueError(f"the engine implements the Qwen2 decoder family only (model_type={mt!r})")
    if c.get("hidden_act", "silu") != "silu":
        raise ValueError(f"unsupported hidden_act {c.get('hidden_act')!r}")
    if c.get("use_sliding_window"):
        raise ValueError("sliding-window attention is not supported")
    out = {k: c[k] for k in _HF_DIM_KEYS if k in c and c[k] is not None}
    rp = c.get("rope_parameters") or c.get("rope_scaling") or {}
    if "rope_theta" not in out and isinstance(rp, dict) and rp.get("rope_theta") is not None:
        out["rope_theta"] = rp["rope_theta"]
    if isinstance(rp, dict) and rp.get("rope_type", rp.get("type", "default")) not in ("default", None):
        raise ValueError(f"unsupported rope scaling {rp}")
    return out


def read_hf_weights(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of a HuggingFace checkpoint directory: `model.safetensors`, the sharded form behind
    `model.safetensors.index.json`, or `pytorch_model.bin`."""
    from safetensors.torch import load_file
    one = os.path.join(path, "model.safetensors")
    idx = one + ".index.json"
    if os.path.exists(one):
        return load_file(one)
    if os.path.exists(idx):
        with open(idx) as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
        sd: Dict[str, torch.Tensor] = {}
        for sh in shards:
            sd.update(load_file(os.path.join(path, sh)))
        return sd
    b = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(b):
        return torch.load(b, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no model.safetensors / model.safetensors.index.json / pytorch_model.bin under {path}")


@dataclass
class UnitLMConfig:
    """unit_lm.py:32-79. `base_config` is a dict of Qwen2Config-named fields (or None to look
    `base_model_name` up in KNOWN_BASE_CONFIGS); extra kwargs such as `rope_theta` override it the
    way the reference forwards **kwargs into AutoConfig.from_pretrained (config/model/slam.yaml:8)."""
    base_model_name: str = "Qwen/Qwen2.5-0.5B"
    base_config: Optional[dict] = None
    vocab_size: int = 502
    twist_init: bool = False
    use_cache: bool = False
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 1
    torch_dtype: Optional[str] = "bfloat16"
    attn_implementation: Optional[str] = "flash_attention_2"  # the engine's attention is varlen-capable
    max_tokens: int = 8192          # engine workspace capacity (B*T per micro-batch)
    extra: dict = field(default_factory=dict)

    def __init__(self, base_model_name="Qwen/Qwen2.5-0.5B", base_config=None, vocab_size=502, twist_init=False,
                 use_cache=False, pad_token_id=0, bos_token_id=1, eos_token_id=1, torch_dtype="bfloat16",
                 attn_implementation="flash_attention_2", max_tokens=8192, **kwargs):
        self.base_model_name = base_model_name
        local_dir = isinstance(base_model_name, str) and os.path.isfile(os.path.join(base_model_name, "config.json"))
        if base_config is None:
            if local_dir:  # "could use a huggingface model name or a path to a model" (unit_lm.py:47)
                with open(os.path.join(base_model_name, "config.json")) as f:
                    base_config = base_config_from_hf(json.load(f))
            elif base_model_name in KNOWN_BASE_CONFIGS:
                base_config = dict(KNOWN_BASE_CONFIGS[base_model_name])
            else:
                raise ValueError(f"unknown base model {base_model_name!r}: pass a local HuggingFace directory or "
                                 f"base_config=dict(...) (no hub access); known: {sorted(KNOWN_BASE_CONFIGS)}")
        elif "model_type" in base_config or "rope_parameters" in base_config:
            base_config = base_config_from_hf(base_config)  # the reference's serialised Qwen2Config
        base_config = dict(base_config)
        for k in list(kwargs):
            if k in ("rope_theta", "rms_norm_eps", "initializer_range"):
                base_config[k] = kwargs.pop(k)
        base_config.setdefault("head_dim", base_config["hidden_size"] // base_config["num_attention_heads"])
        base_config.setdefault("rms_norm_eps", 1e-6)
        base_config.setdefault("rope_theta", 10000.0)
        base_config.setdefault("initializer_range", 0.02)
        base_config.setdefault("tie_word_embeddings", True)
        base_config.update(pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id)
        if not base_config["tie_word_embeddings"]:
            raise ValueError("the engine supports tied embeddings only (Slam / Qwen2.5-0.5B)")
        self.base_config = base_config
        self.vocab_size = vocab_size
        self.twist_init = twist_init
        if twist_init and not local_dir:
            raise ValueError(f"twist_init=True loads the text LM's weights (unit_lm.py:94-98): base_model_name must be a "
                             f"local HuggingFace checkpoint directory (the hub is unreachable), got {base_model_name!r}")
        self.use_cache = use_cache
        self.pad_token_id, self.bos_token_id, self.eos_token_id = pad_token_id, bos_token_id, eos_token_id
        self.torch_dtype = torch_dtype
        self.attn_implementation = attn_implementation
        self._attn_implementation = attn_implementation  # read at cli/train.py:43
        self.max_tokens = max_tokens
        self.tie_word_embeddings = True
        self.extra = kwargs

    def to_dict(self):
        return dict(model_type="speech_language_model", engine="slamkit_amd", base_model_name=self.base_model_name,
                    base_config=self.base_config, vocab_size=self.vocab_size, twist_init=False, pad_token_id=self.pad_token_id,
                    bos_token_id=self.bos_token_id, eos_token_id=self.eos_token_id, torch_dtype=self.torch_dtype,
                    max_tokens=self.max_tokens)

    def engine_desc(self) -> E.SlamModelDesc:
        b = self.base_config
        return E.SlamModelDesc(b["num_hidden_layers"], b["hidden_size"], b["num_attention_heads"],
                               b["num_key_value_heads"], b["head_dim"], b["intermediate_size"], self.vocab_size,
                               self.pad_token_id if self.pad_token_id is not None else -1,
                               float(b["rms_norm_eps"]), float(b["rope_theta"]))


@dataclass
class CausalLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        return getattr(self, k)


class _EngineLoss(torch.autograd.Function):
    """Lets `out.loss.backward()` drive slam_backward (plugin-surface compatibility). The trainer's
    hot loop calls UnitLM.backward() directly and avoids the host read of grad_output."""

    @staticmethod
    def forward(ctx, anchor, model, loss):
        ctx.model = model
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.model.backward(float(grad_out))
        return torch.zeros(1, device=grad_out.device), None, None


def _right_padded(mask: torch.Tensor) -> bool:
    m = mask.to(torch.int64)
    return bool((m[:, 1:] <= m[:, :-1]).all())


class UnitLM(TokenLM):
    """unit_lm.py:82-212 on the HIP engine."""
    base_model_prefix = "lm"

    def __init__(self, config: UnitLMConfig, device: Optional[str] = None, seed: int = 0, allocate_grads: bool = True,
                 _from_pretrained: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("slamkit_amd.UnitLM needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.config = config
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.engine = E.Engine(config.engine_desc())
        n = self.engine.n_params
        with torch.cuda.device(self.device):
            self.flat_params = torch.zeros(n, dtype=torch.bfloat16, device=self.device)
            self.flat_master = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.flat_grads = torch.zeros(n, dtype=torch.float32, device=self.device) if allocate_grads else None
            self.engine.bind_params(self.flat_params, self.flat_grads)
            b = config.base_config
            dims = (b["hidden_size"], b["intermediate_size"], b["num_attention_heads"] * b["head_dim"],
                    (b["num_attention_heads"] + 2 * b["num_key_value_heads"]) * b["head_dim"])
            self.flat_params_t = None
            if allocate_grads and all(x % 64 == 0 for x in dims):
                # transposed weight images so dgrad runs on the LDS-DMA GEMM path (training only)
                self.flat_params_t = torch.zeros(n, dtype=torch.bfloat16, device=self.device)
                self.engine.bind_params_t(self.flat_params_t)
            self._ws = None
            self._ws_tokens = 0
            self._ensure_workspace(config.max_tokens)
            self._loss_buf = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
            self.flat_grads16 = None       # bf16 gradients of the step's last backward (enable_bf16_grads)
            self._grads_in_bf16 = False    # the last backward left its final values there, not in flat_grads
        for opt in ("fuse_swiglu", "fuse_dswiglu", "gemm_group_rows", "gemm_glds", "gemm_tn_splits", "gemm_tn_balanced", "gemm_256", "gemm_nt224", "gemm_nt224_min_k", "gemm_256_dswiglu", "gemm_256_persist", "gemm_256_stagger", "gemm_256_stagger_dswiglu", "gemm_256_cohorts", "gemm_256_persist_cus", "gemm_group_cols_256", "gemm_group_rows_256", "gemm_mf32", "gemm_256_w4", "gemm_256_roles", "gemm_256_batch_loads", "gemm_tn224", "gemm_tn224_min_m", "gemm_tn224_max_split", "gemm_tn_bal_bg_max_split", "gemm_tn224_bg_min_m", "gemm_tn224_bg_max_split", "bwd_wgrad_stream", "bwd_aux_side", "bwd_wgrad_cus", "attn_jq", "attn_kw", "attn_nch", "attn_prio", "fuse_adamw_t"):  # tuning overrides, e.g. SLAM_FUSE_SWIGLU=0
            v = os.environ.get("SLAM_" + opt.upper())
            if v is not None:
                self.engine.set_option(opt, int(v))
        self.training = True
        self._build_key_map()
        self.init_weights(seed)
        if config.twist_init and not _from_pretrained:
            # TWIST: start from the text LM's weights, keep the first vocab_size embedding rows (unit_lm.py:94-102)
            self.load_hf_text_lm(config.base_model_name)

    # ---- layout ------------------------------------------------------------------------------
    def _build_key_map(self):
        b, t = self.config.base_config, self.engine.tensors
        nH, nKV, hd, I = b["num_attention_heads"], b["num_key_value_heads"], b["head_dim"], b["intermediate_size"]
        H = b["hidden_size"]
        km: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        km["lm.model.embed_tokens.weight"] = (t["embed"].offset, (self.config.vocab_size, H))
        for l in range(b["num_hidden_layers"]):
            p, q = f"lm.model.layers.{l}.", f"layers.{l}."
            o = t[q + "wqkv"].offset
            km[p + "self_attn.q_proj.weight"] = (o, (nH * hd, H))
            km[p + "self_attn.k_proj.weight"] = (o + nH * hd * H, (nKV * hd, H))
            km[p + "self_attn.v_proj.weight"] = (o + (nH + nKV) * hd * H, (nKV * hd, H))
            o = t[q + "bqkv"].offset
            km[p + "self_attn.q_proj.bias"] = (o, (nH * hd,))
            km[p + "self_attn.k_proj.bias"] = (o + nH * hd, (nKV * hd,))
            km[p + "self_attn.v_proj.bias"] = (o + (nH + nKV) * hd, (nKV * hd,))
            km[p + "self_attn.o_proj.weight"] = (t[q + "wo"].offset, (H, nH * hd))
            o = t[q + "wgu"].offset  # rows interleaved in blocks of 32 gate / 32 up (slam_engine.h)
            km[p + "mlp.gate_proj.weight"] = (o, (I, H), 0)
            km[p + "mlp.up_proj.weight"] = (o, (I, H), 1)
            km[p + "mlp.down_proj.weight"] = (t[q + "wd"].offset, (H, I))
            km[p + "input_layernorm.weight"] = (t[q + "ln1"].offset, (H,))
            km[p + "post_attention_layernorm.weight"] = (t[q + "ln2"].offset, (H,))
        km["lm.model.norm.weight"] = (t["norm"].offset, (H,))
        self.key_map = km

    def _view(self, flat: torch.Tensor, key: str, writable: bool = False) -> torch.Tensor:
        """HF-named window of a flat engine buffer. gate_proj / up_proj live interleaved in 32-row blocks:
        `writable` returns the strided [I/32, 32, H] view (copy_ from src.view(I//32, 32, H)), otherwise
        an [I, H] copy."""
        ent = self.key_map[key]
        off, shp = ent[0], ent[1]
        if len(ent) == 3:
            I, H = shp
            v = flat[off:off + 2 * I * H].view(I // 32, 2, 32, H)[:, ent[2]]
            return v if writable else v.reshape(I, H)
        n = 1
        for s in shp:
            n *= s
        return flat[off:off + n].view(*shp)

    def _assign(self, flat: torch.Tensor, key: str, src: torch.Tensor):
        v = self._view(flat, key, writable=True)
        v.copy_(src.to(device=flat.device, dtype=flat.dtype).reshape(v.shape))

    def _ensure_workspace(self, tokens: int):
        if tokens <= self._ws_tokens:
            return
        nbytes = self.engine.workspace_bytes(tokens)
        self._ws = None
        self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self._ws.data_ptr()) % 256
        self._ws_aligned = self._ws[off:off + nbytes]
        self.engine.bind_workspace(self._ws_aligned, tokens)
        self._ws_tokens = tokens

    # ---- parameters ----------------------------------------------------------------------------
    @torch.no_grad()
    def init_weights(self, seed: int = 0):
        """HF `_init_weights`: N(0, initializer_range) matrices and embeddings, zero biases, unit
        norms, zero padding_idx row (unit_lm.py:114-115 -> transformers PreTrainedModel)."""
        std = float(self.config.base_config["initializer_range"])
        g = torch.Generator(device=self.device).manual_seed(seed)
        self.engine.join()
        self._weights.zero_()
        for k in self.key_map:
            v = self._view(self._weights, k, writable=True)
            if k.endswith("norm.weight"):
                v.fill_(1.0)
            elif k.endswith(".bias"):
                v.zero_()
            else:
                v.normal_(0.0, std, generator=g)
        if self.config.pad_token_id is not None and self.config.pad_token_id >= 0:
            self._view(self._weights, "lm.model.embed_tokens.weight", True)[self.config.pad_token_id].zero_()
        self.sync_params_from_master()

    def sync_params_from_master(self):
        if self.flat_master is not None:
            self.engine.cast_params(self.flat_master)
        else:
            self.engine.refresh_transposed()

    def drop_master(self):
        """bf16-parameter training (the recipe's precision, slam.yaml:9): the bf16 buffer becomes the only copy of the
        weights; the optimizer (slam_adamw_step_bf16) updates it in place."""
        self.engine.join()
        self.flat_master = None

    @property
    def _weights(self) -> torch.Tensor:
        """The authoritative flat weight buffer: fp32 master when there is one, else the bf16 parameters."""
        return self.flat_master if self.flat_master is not None else self.flat_params

    def named_parameters(self) -> Iterator[Tuple[str, torch.Tensor]]:
        self.engine.join()  # a pending overlapped optimizer step writes these buffers on the engine's side stream
        for k in self.key_map:
            yield k, self._view(self.flat_params, k)

    def parameters(self) -> Iterator[torch.Tensor]:
        for _, v in self.named_parameters():
            yield v

    def named_grads(self) -> Iterator[Tuple[str, torch.Tensor]]:
        """(name, gradient) of the last backward: views of the fp32 buffer, or - when that backward kept its final values in
        bf16 only (`backward(final=2)`: the reference's own gradient precision) - fp32 copies of the bf16 buffer's windows."""
        self.engine.join()
        for k in self.key_map:
            if self._grads_in_bf16:
                yield k, self._view(self.flat_grads16, k).float()
            else:
                yield k, self._view(self.flat_grads, k)

    def enable_bf16_grads(self):
        """Allocate the bf16 gradient buffer `backward(final=2)` stores the step's final gradient values in."""
        if self.flat_grads16 is None:
            with torch.cuda.device(self.device):
                self.flat_grads16 = torch.zeros(self.engine.n_params, dtype=torch.bfloat16, device=self.device)
        return self.flat_grads16

    def num_parameters(self) -> int:
        return sum(v.numel() for v in self.parameters())

    def state_dict(self, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
        self.engine.join()
        src = self._weights if dtype == torch.float32 else self.flat_params
        return {k: self._view(src, k).detach().to(dtype).cpu().clone() for k in self.key_map}

    @staticmethod
    def _canonical_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Key layouts accepted: the UnitLM layout `lm.model.*` (reference and engine checkpoints) and a raw
        HuggingFace causal LM's `model.*` / `lm_head.weight` (text-LM weights for TWIST initialisation)."""
        if any(k.startswith("lm.") for k in sd):
            return dict(sd)
        return {("lm." + k if k.startswith(("model.", "lm_head.")) else k): v for k, v in sd.items()}

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Copies the tensors into the fp32 master and refreshes the bf16 images. The tied `lm_head.weight` is accepted and
        ignored. The embedding follows `resize_token_embeddings` (unit_lm.py:102): a longer matrix is cut to the first
        vocab_size rows; a shorter one fills the new rows with the mean of the old rows (transformers' mean-resizing
        draws them from N(mean, 1e-9 * cov): within 1e-5 of the mean). Returns (missing, unexpected); with `strict`
        either being non-empty raises."""
        self.engine.join()
        sd = self._canonical_keys(sd)
        missing = [k for k in self.key_map if k not in sd]
        extra = [k for k in sd if k not in self.key_map and not k.endswith("lm_head.weight")]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: {len(missing)} missing (first: {missing[:4]}), "
                           f"{len(extra)} unexpected (first: {extra[:4]})")
        for k in self.key_map:
            if k in sd:
                src = sd[k]
                shp = self.key_map[k][1]
                if k == "lm.model.embed_tokens.weight" and src.shape[0] != shp[0]:
                    if src.shape[0] > shp[0]:
                        src = src[:shp[0]]
                    else:
                        src = torch.cat([src.float(), src.float().mean(0, keepdim=True).expand(shp[0] - src.shape[0], -1)])
                if tuple(src.shape) != tuple(shp):
                    raise ValueError(f"{k}: checkpoint shape {tuple(src.shape)} != model shape {tuple(shp)}")
                self._assign(self._weights, k, src)
        self.sync_params_from_master()
        return missing, extra

    def load_hf_text_lm(self, path: str):
        """TWIST initialisation (unit_lm.py:94-102): every weight of a local HuggingFace Qwen2 text LM, embedding rows
        resized to vocab_size. The text LM must tie its head (an untied `lm_head` would be dropped silently)."""
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        if not c.get("tie_word_embeddings", True):
            raise ValueError("the engine supports tied embeddings only; this text LM has an untied lm_head")
        want = base_config_from_hf(c)
        for k in ("num_hidden_layers", "hidden_size", "num_attention_heads", "num_key_value_heads", "intermediate_size"):
            if want.get(k) != self.config.base_config.get(k):
                raise ValueError(f"text LM {k}={want.get(k)} != model {k}={self.config.base_config.get(k)}")
        self.load_state_dict(read_hf_weights(path), strict=True)
        return self

    def get_input_embeddings(self):
        return self._view(self.flat_params, "lm.model.embed_tokens.weight")

    def get_output_embeddings(self):
        return self.get_input_embeddings()  # tied

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    # ---- forward / backward ---------------------------------------------------------------------
    def _segments(self, position_ids: torch.Tensor):
        """Packed [1, sum T] batches (DataCollatorWithFlattening): per-token sequence bounds from
        position_ids == 0 restarts."""
        pos = position_ids.reshape(-1)
        M = pos.numel()
        idx = torch.arange(M, device=pos.device, dtype=torch.int64)
        is0 = pos == 0
        start = torch.cummax(torch.where(is0, idx, torch.zeros_like(idx)), 0).values
        nxt = torch.where(is0, idx, torch.full_like(idx, M))
        # end = position of the next restart strictly after the token
        nxt_shift = torch.cat([nxt[1:], torch.full((1,), M, device=pos.device, dtype=torch.int64)])
        end = torch.flip(torch.cummin(torch.flip(nxt_shift, [0]), 0).values, [0])
        return start.to(torch.int32).contiguous(), end.to(torch.int32).contiguous()

    def forward(self, input_ids: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                num_items_in_batch=None, return_logits: bool = True, **unused) -> CausalLMOutput:
        """UnitLM.forward (unit_lm.py:135-182). `attention_mask` must be right padding (what
        DataCollatorForLanguageModeling produces): under the causal mask it never changes a real
        token's output, so the engine does not read it."""
        assert input_ids is not None and input_ids.dim() == 2
        B, T = input_ids.shape
        if attention_mask is not None and not attention_mask.is_cuda and not _right_padded(attention_mask):
            # collated batches arrive on the host (the trainer's boundary): checked there. A DEVICE mask is not read back -
            # that would stall the host behind the previous step's kernels in the hot loop; callers that build masks on
            # the device (synthetic benches, DPO's padded pairs) construct right padding by definition.
            raise ValueError("only right-padded attention_mask is supported")
        if position_ids is not None and B > 1 and not position_ids.is_cuda:
            # [B > 1, T] with positions is the dense layout (every row 0..T-1); a packed batch mis-shaped as several rows
            # would silently lose its segment bounds (the engine takes segments from a [1, sum T] row only)
            if not bool((position_ids == torch.arange(T, dtype=position_ids.dtype)[None]).all()):
                raise ValueError("position_ids with batch size > 1 must be plain aranges; packed batches are [1, sum T]")
        dev = self.device
        nb = not input_ids.is_cuda and input_ids.is_pinned()  # pinned host batches (the trainer's prefetch thread): async H2D
        ids = input_ids.to(dev, torch.int64, non_blocking=nb)
        lab = labels.to(dev, torch.int64, non_blocking=nb) if labels is not None else None
        pos = position_ids.to(dev, torch.int64, non_blocking=nb) if position_ids is not None else None
        # The LDS-DMA wgrad path needs a token count that is a multiple of 64; collated batches have arbitrary lengths.
        # Right-pad the token axis with pad ids / ignored labels (a dummy trailing segment for packed rows): under the
        # causal mask no real token sees the padding, the loss skips it, and the extra logits rows are not returned.
        T0 = T
        if (B * T) % 64 and (pos is None or B == 1):
            T = -(-T // 64) * 64
            extra = T - T0
            ids = torch.cat([ids, ids.new_full((B, extra), int(self.config.pad_token_id or 0))], 1)
            if lab is not None:
                lab = torch.cat([lab, lab.new_full((B, extra), -100)], 1)
            if pos is not None:
                pos = torch.cat([pos, torch.arange(extra, device=dev, dtype=torch.int64)[None]], 1)
        ids = ids.contiguous()
        lab = lab.contiguous() if lab is not None else None
        self._ensure_workspace(B * T)
        seg_s = seg_e = None
        if pos is not None:
            pos = pos.contiguous()
            if B == 1:
                seg_s, seg_e = self._segments(pos)
        logits = torch.empty(B, T, self.config.vocab_size, dtype=torch.bfloat16, device=dev) if return_logits else None
        if isinstance(num_items_in_batch, torch.Tensor):
            num_items_in_batch = float(num_items_in_batch)
        self._hold = (ids, lab, pos, seg_s, seg_e)  # the engine borrows these until backward
        self.engine.forward(ids, lab, pos, seg_s, seg_e, B, T,
                            float(num_items_in_batch) if num_items_in_batch else 0.0,
                            self._loss_buf if lab is not None else None, logits)
        loss = None
        if lab is not None:
            loss = _EngineLoss.apply(self._anchor, self, self._loss_buf) if torch.is_grad_enabled() else self._loss_buf.clone()
        if logits is not None and T != T0:
            logits = logits[:, :T0]
        return CausalLMOutput(loss=loss, logits=logits)

    __call__ = forward

    def backward(self, grad_scale: float = 1.0, bucket_layers: int = 0, bucket_cb=None, final: int = 0):
        """d(loss * grad_scale)/dparams accumulated into `flat_grads` (fp32). final = 1 | 2 marks the last backward of an
        optimizer step (engine option "grad_final_next"): the gradient-norm partials come out of the final-value stores, and
        with 2 the final values are kept in bf16 only (`flat_grads16`; `named_grads` reads them there). With a bucket callback
        (data parallel) `flat_grads16` is also the buffer that crosses the wire: the exchanged gradients stay in it."""
        if final == 2:
            self.engine.set_grad_image(self.enable_bf16_grads())
        self.engine.backward(grad_scale, bucket_layers, bucket_cb, final=final)
        self._grads_in_bf16 = final == 2

    def zero_grad(self):
        self.engine.zero_grads()

    @torch.no_grad()
    def log_likelihood(self, tokens: torch.Tensor, mean_nll: bool, ignore_tokens: Optional[List[int]] = None) -> torch.Tensor:
        """unit_lm.py:184-194 + calc_nll (calculation_utils.py:5-29): pad -> -100, per-sequence
        sum (or mean) of target log-probs."""
        B, T = tokens.shape
        ids = tokens.to(self.device, torch.int64).contiguous()
        lab = ids.clone()
        lab[lab == self.config.pad_token_id] = -100
        self._ensure_workspace(B * T)
        self._hold = (ids, lab)
        mask = None
        if ignore_tokens is not None:  # logits[:, :, ignore_tokens] = -inf (unit_lm.py:187-188), done inside the CE kernel
            mask = torch.zeros(self.engine.padded_vocab(), dtype=torch.uint8, device=self.device)
            mask[torch.as_tensor(list(ignore_tokens), dtype=torch.long, device=self.device)] = 1
            self.engine.set_logit_mask(mask)
        try:
            self.engine.forward(ids, lab, None, None, None, B, T, 0.0, self._loss_buf, None)
        finally:
            if mask is not None:
                torch.cuda.current_stream(self.device).synchronize()  # the kernel reads the mask: keep it alive until done
                self.engine.set_logit_mask(None)
        ll = torch.empty(B, dtype=torch.float32, device=self.device)
        cnt = torch.empty(B, dtype=torch.float32, device=self.device)
        self.engine.seq_loglik(lab, B, T, ll, cnt)
        return ll / cnt if mean_nll else ll

    def sequence_logps(self, input_ids: torch.Tensor, labels: torch.Tensor):
        """Per-sequence sums of target log-probs over the non-ignored labels (what TRL's DPOTrainer calls
        `chosen_logps` / `rejected_logps`). Leaves the engine ready for `scale_loss_rows` + `backward`:
        d(-logp_b)/dlogits is stored unscaled (num_items = 1)."""
        B, T = input_ids.shape
        ids = input_ids.to(self.device, torch.int64).contiguous()
        lab = labels.to(self.device, torch.int64).contiguous()
        self._ensure_workspace(B * T)
        self._hold = (ids, lab)
        self.engine.forward(ids, lab, None, None, None, B, T, 1.0, self._loss_buf, None)
        ll = torch.empty(B, dtype=torch.float32, device=self.device)
        cnt = torch.empty(B, dtype=torch.float32, device=self.device)
        self.engine.seq_loglik(lab, B, T, ll, cnt)
        return ll, cnt

    def backward_sequence_loss(self, seq_coef: torch.Tensor, B: int, T: int, grad_scale: float = 1.0, **kw):
        """Backward of sum_b seq_coef[b] * (-logp_b): seq_coef = d loss / d(-logp_b)."""
        coef = seq_coef.to(self.device, torch.float32).contiguous()
        self._hold = self._hold + (coef,)
        self.engine.scale_loss_rows(coef, B, T)
        self.engine.backward(grad_scale, kw.get("bucket_layers", 0), kw.get("bucket_cb"))
        self._grads_in_bf16 = False

    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, generation_config=None, max_new_tokens: int = 32,
                 do_sample: bool = False, temperature: float = 1.0, top_k: int = 0, seed: Optional[int] = None,
                 **kwargs) -> torch.Tensor:
        """Minimal sampler for the TokenLM surface (unit_lm.py:196-198): full re-forward per token
        (no KV cache - evaluation/generation is outside the training hot path)."""
        if generation_config is not None:
            max_new_tokens = getattr(generation_config, "max_new_tokens", max_new_tokens) or max_new_tokens
            do_sample = getattr(generation_config, "do_sample", do_sample)
        seq = inputs.to(self.device, torch.int64)
        g = torch.Generator(device=self.device)
        if seed is not None:
            g.manual_seed(seed)
        done = torch.zeros(seq.shape[0], dtype=torch.bool, device=self.device)
        for _ in range(max_new_tokens):
            logits = self.forward(seq).logits[:, -1].float()
            if do_sample:
                logits = logits / max(temperature, 1e-6)
                if top_k:
                    kth = torch.topk(logits, top_k).values[:, -1:]
                    logits = logits.masked_fill(logits < kth, float("-inf"))
                nxt = torch.multinomial(torch.softmax(logits, -1), 1, generator=g)[:, 0]
            else:
                nxt = logits.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, self.config.pad_token_id), nxt)
            seq = torch.cat([seq, nxt[:, None]], 1)
            done |= nxt == self.config.eos_token_id
            if bool(done.all()):
                break
        return seq

    # ---- checkpoints (HF layout) ------------------------------------------------------------------
    def save_pretrained(self, save_directory: str, dtype=torch.bfloat16):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        sd = self.state_dict(dtype)
        save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=1)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, **kwargs) -> "UnitLM":
        """Loads a checkpoint directory written by this engine, by the reference's `UnitLM.save_pretrained`
        (unit_lm.py:200-212: `config.json` with a serialised Qwen2Config under `base_config`, weights under `lm.model.*`)
        or - as a convenience for converted text LMs - a raw HuggingFace Qwen2 directory (`model.*` keys; pass
        vocab_size= to resize). Every parameter must be present in the file: a mismatching layout raises instead of
        leaving the model at its random initialisation."""
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        vocab = kwargs.pop("vocab_size", None)
        if "base_config" in c:
            base = c["base_config"]
            name = c.get("base_model_name", "local")
        elif c.get("model_type") == "qwen2":
            base, name = c, path
        else:
            raise ValueError(f"{path}/config.json is neither a UnitLM config (no `base_config`) nor a Qwen2 config "
                             f"(model_type={c.get('model_type')!r})")
        if not (isinstance(name, str) and (name in KNOWN_BASE_CONFIGS or os.path.isdir(name))):
            name = "local"  # e.g. a hub id or a path of the machine that wrote the checkpoint: the dims are in base_config
        cfg = UnitLMConfig(base_model_name=name, base_config=base,
                           vocab_size=vocab if vocab is not None else c["vocab_size"],
                           pad_token_id=c.get("pad_token_id", base.get("pad_token_id", 0)),
                           bos_token_id=c.get("bos_token_id", base.get("bos_token_id", 1)),
                           eos_token_id=c.get("eos_token_id", base.get("eos_token_id", 1)),
                           max_tokens=kwargs.pop("max_tokens", c.get("max_tokens", 8192)))
        m = cls(cfg, _from_pretrained=True, **kwargs)
        m.load_state_dict(read_hf_weights(path), strict=True)
        return m
