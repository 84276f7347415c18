import os, sys, math, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from oracle import slam_oracle as O
from slamkit_amd.model import UnitLM, UnitLMConfig
from tests.test_gpu_model import _packed_row
from tests.gpu_util import rel_err
L = int(os.environ.get("L", "2"))
cfg = O.OracleConfig(n_layers=L, hidden=1536, n_heads=12, n_kv_heads=2, head_dim=128, intermediate=8960, vocab=152167, rope_theta=1000000.0)
base = dict(num_hidden_layers=cfg.n_layers, hidden_size=cfg.hidden, num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads,
            head_dim=cfg.head_dim, intermediate_size=cfg.intermediate, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=True)
m = UnitLM(UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, max_tokens=16384), seed=1)
gen = torch.Generator().manual_seed(99)
lens16 = [2048, 1531, 64, 2000, 777, 2048, 1200, 300, 1900, 2048, 468]
lens16.append(16384 - sum(lens16))
seqs = [_packed_row([n], cfg.vocab, 151667, gen) for n in lens16]
def run(order):
    i_, p_, l_ = (torch.cat([seqs[j][k] for j in order], 1) for k in range(3))
    m.zero_grad()
    o = m(input_ids=i_, position_ids=p_, labels=l_, return_logits=False)
    m.backward(); torch.cuda.synchronize()
    return float(o.loss), {k: v.clone() for k, v in m.named_grads()}
for opt in [{}, {"gemm_tn224": 0}, {"bwd_wgrad_stream": 0}]:
    for k, v in opt.items(): m.engine.set_option(k, v)
    l0, g0 = run(range(12)); l2, g2 = run([3, 0, 11, 7, 1, 9, 2, 10, 5, 4, 8, 6])
    worst = sorted(((rel_err(g2[k], g0[k]), k) for k in g0), reverse=True)[:6]
    print(opt, "loss", l0, l2, "worst per-tensor rel_err:", [(round(a, 4), k.replace("lm.model.", "")) for a, k in worst], flush=True)
    for k, v in opt.items(): m.engine.set_option(k, 1)
