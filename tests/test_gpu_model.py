"""-m gpu end-to-end parity: the engine-backed UnitLM against (a) golden vectors produced by the real
reference (tests/golden, fp32 HF path) and (b) the CPU oracle on the same bf16-rounded weights.

Stated tolerances (bf16 engine vs fp32 oracle): loss abs <= 2e-2, logits rel-RMS
<= 2e-2, per-tensor gradient cosine >= 0.999 (>= 0.99 for the tiny-norm bias / layernorm vectors)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import slam_oracle as O
from tests.gpu_util import check, cosine, rel_err

pytestmark = pytest.mark.gpu


def _mk(cfg: O.OracleConfig, sd, max_tokens=4096):
    from slamkit_amd.model import UnitLM, UnitLMConfig
    base = dict(num_hidden_layers=cfg.n_layers, hidden_size=cfg.hidden, num_attention_heads=cfg.n_heads,
                num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate,
                rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=True)
    m = UnitLM(UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, max_tokens=max_tokens))
    if sd is not None:
        m.load_state_dict(sd)
    return m


@pytest.fixture(scope="module")
def tiny(golden_data):
    meta = golden_data["meta"]
    cfg = O.OracleConfig(**meta["config"])
    sd = O.init_weights(cfg, seed=meta["seed"], bias_std=meta["bias_std"], norm_jitter=meta["norm_jitter"])
    sd_bf = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    return cfg, sd, sd_bf, _mk(cfg, sd)


def test_state_dict_roundtrip(tiny):
    cfg, sd, sd_bf, m = tiny
    out = m.state_dict(torch.float32)
    assert set(out) == set(sd)
    for k in sd:
        assert torch.equal(out[k], sd[k]), k
    outb = m.state_dict(torch.bfloat16)
    for k in sd:
        assert torch.equal(outb[k].float(), sd_bf[k]), k
    assert m.num_parameters() == sum(v.numel() for v in sd.values())


def test_padded_batch_vs_reference_golden(tiny, golden_npz):
    cfg, sd, sd_bf, m = tiny
    g = golden_npz
    ids, am, lab = (torch.from_numpy(g[k]) for k in ("pad_ids", "pad_mask", "pad_labels"))
    out = m(input_ids=ids, attention_mask=am, labels=lab)
    valid = am.bool()
    check("logits vs reference golden (fp32 HF)", out.logits.float().cpu()[valid], torch.from_numpy(g["pad_logits"])[valid], 2e-2)
    assert abs(float(out.loss) - float(g["pad_loss_mean"])) <= 2e-2
    n = int(g["pad_num_items"])
    out2 = m(input_ids=ids, attention_mask=am, labels=lab, num_items_in_batch=n)
    assert abs(float(out2.loss) - float(g["pad_loss_sum"])) <= 2e-2
    print("loss engine/ref mean:", float(out.loss), float(g["pad_loss_mean"]), "sum:", float(out2.loss), float(g["pad_loss_sum"]))


def test_padded_batch_grads_vs_oracle(tiny, golden_npz):
    cfg, sd, sd_bf, m = tiny
    g = golden_npz
    ids, am, lab = (torch.from_numpy(g[k]) for k in ("pad_ids", "pad_mask", "pad_labels"))
    loss_ref, logits_ref, grads_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, attention_mask=am)
    m.zero_grad()
    out = m(input_ids=ids, attention_mask=am, labels=lab)
    out.loss.backward()  # plugin-surface path through autograd
    torch.cuda.synchronize()
    assert abs(float(out.loss) - float(loss_ref)) <= 5e-3
    check("logits vs oracle (same bf16 weights)", out.logits.float().cpu()[am.bool()], logits_ref[am.bool()], 1e-2)
    worst = 1.0
    for k, gv in m.named_grads():
        ref = grads_ref[k]
        c = cosine(gv, ref)
        r = rel_err(gv, ref)
        worst = min(worst, c)
        small = k.endswith(".bias") or k.endswith("norm.weight")
        assert c >= (0.99 if small else 0.999), f"{k}: cosine {c:.5f} rel {r:.3e}"
        assert abs(float(gv.norm()) / float(ref.norm()) - 1) <= 3e-2, k
    print("worst grad cosine", worst)
    # against the reference-produced full gradients too (fp32 weights there -> looser)
    for key in [k for k in g if k.startswith("pad_gradfull/")]:
        name = key.split("/", 1)[1]
        c = cosine(dict(m.named_grads())[name], torch.from_numpy(g[key]))
        assert c >= 0.99, f"{name} vs reference golden: cosine {c:.5f}"
    # padding_idx: row 0 only gets the tied-head contribution; pad rows 502..511 of the image stay 0
    E = m.flat_grads[: 512 * cfg.hidden].view(512, cfg.hidden)
    assert float(E[cfg.vocab:].abs().max()) == 0.0


def test_backward_accumulates_and_scales(tiny, golden_npz):
    cfg, sd, sd_bf, m = tiny
    ids, am, lab = (torch.from_numpy(golden_npz[k]) for k in ("pad_ids", "pad_mask", "pad_labels"))
    m.zero_grad()
    m(input_ids=ids, labels=lab, return_logits=False)
    m.backward(1.0)
    g1 = m.flat_grads.clone()
    m(input_ids=ids, labels=lab, return_logits=False)
    m.backward(2.0)
    g3 = m.flat_grads.clone()
    torch.cuda.synchronize()
    assert rel_err(g3, 3 * g1) <= 1e-2  # grad accumulation (GA) + loss scaling, bf16 dlogits rounding only
    # determinism: same inputs -> bit-identical gradients (no atomics anywhere)
    m.zero_grad()
    m(input_ids=ids, labels=lab, return_logits=False)
    m.backward(1.0)
    assert torch.equal(m.flat_grads, g1)


def test_packed_batch_vs_reference_golden(tiny, golden_npz):
    cfg, sd, sd_bf, m = tiny
    g = golden_npz
    ids, pos, lab = (torch.from_numpy(g[k]) for k in ("pack_ids", "pack_pos", "pack_labels"))
    out = m(input_ids=ids, position_ids=pos, labels=lab)
    check("packed logits vs per-sequence reference", out.logits.float().cpu(), torch.from_numpy(g["pack_logits"]), 2e-2)
    assert abs(float(out.loss) - float(g["pack_loss_mean"])) <= 2e-2
    # gradients of the packed path vs the oracle
    loss_ref, _, grads_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, position_ids=pos, packed=True)
    m.zero_grad()
    m(input_ids=ids, position_ids=pos, labels=lab, return_logits=False)
    m.backward()
    for k, gv in m.named_grads():
        small = k.endswith(".bias") or k.endswith("norm.weight")
        assert cosine(gv, grads_ref[k]) >= (0.99 if small else 0.999), k


def test_log_likelihood_vs_reference_golden(tiny, golden_npz):
    cfg, sd, sd_bf, m = tiny
    ids = torch.from_numpy(golden_npz["pad_ids"])
    for mean_nll, key in ((True, "ll_mean"), (False, "ll_sum")):
        ll = m.log_likelihood(ids, mean_nll).cpu().numpy()
        ref = golden_npz[key]
        assert np.allclose(ll, ref, rtol=5e-3, atol=2e-2), (key, ll, ref)
    # ignore_tokens (one-wave CE kernel, V <= 512) against the oracle on the same bf16 weights
    present = set(ids.flatten().tolist())
    ignore = [t for t in range(50, cfg.vocab) if t not in present][:200]
    got = m.log_likelihood(ids, False, ignore_tokens=ignore).cpu().numpy()
    ref = O.log_likelihood(cfg, sd_bf, ids.clone(), False, ignore_tokens=ignore).numpy()
    assert np.allclose(got, ref, rtol=5e-3, atol=0.3), (got, ref)
    # a target inside the ignored set has zero probability: -inf log-likelihood, like the reference
    bad = m.log_likelihood(ids, False, ignore_tokens=[int(ids[0, 3])]).cpu().numpy()
    assert np.isneginf(bad[0])


def test_backward_bucket_ranges_tile_the_flat_gradient(tiny, golden_npz):
    """slam_bucket_cb ranges: disjoint, reported top-down, covering [0, n_params); the last (exposed) one only
    holds layer 0 and the embedding."""
    cfg, sd, sd_bf, m = tiny
    ids, lab = (torch.from_numpy(golden_npz[k]) for k in ("pad_ids", "pad_labels"))
    got = []
    m.zero_grad()
    m(input_ids=ids, labels=lab, return_logits=False)
    m.backward(1.0, 1, lambda off, cnt, stream=None: got.append((off, cnt)))
    torch.cuda.synchronize()
    n = m.engine.n_params
    assert got[0][0] + got[0][1] == n and got[-1][0] == 0
    for (o1, c1), (o2, c2) in zip(got, got[1:]):
        assert o2 + c2 == o1 and c1 > 0 and c2 > 0
    t = m.engine.tensors
    assert got[-1][1] == t["layers.1.ln1"].offset  # embedding + layer 0



def test_bucket_ranges_are_final_on_the_reported_stream(tiny, golden_npz):
    """slam_bucket_stream: a consumer that orders itself ONLY after the stream reported with a bucket (the engine's
    weight-gradient stream for the intermediate buckets, the backward stream for the last) reads the final gradient of
    that range - the contract the data-parallel reducer relies on, with main never waiting at a boundary."""
    cfg, sd, sd_bf, m = tiny
    ids, lab = (torch.from_numpy(golden_npz[k]) for k in ("pad_ids", "pad_labels"))
    consumer = torch.cuda.Stream()
    for two in (1, 0):
        m.engine.set_option("bwd_wgrad_stream", two)
        snaps, streams = [], []

        def cb(off, cnt, stream=None):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream())
            consumer.wait_event(ev)
            with torch.cuda.stream(consumer):
                snaps.append((off, cnt, m.flat_grads[off:off + cnt].clone()))
            streams.append(stream)
        for rep in range(2):
            m.zero_grad()
            m(input_ids=ids, labels=lab, return_logits=False)
            m.backward(1.0, 1, cb)
        torch.cuda.synchronize()
        assert streams[-1] is None and (all(s for s in streams[:len(streams) // 2 - 1]) if two else not any(streams))
        for off, cnt, snap in snaps[len(snaps) // 2:]:
            assert torch.equal(snap, m.flat_grads[off:off + cnt]), (two, off, cnt)
    m.engine.set_option("bwd_wgrad_stream", 1)


def test_wgrad_side_stream_is_bit_identical(tiny, golden_npz):
    """bwd_wgrad_stream: the weight-gradient GEMMs on the engine's second stream (event-ordered against the dgrad chain)
    give the same gradient bits as the single-stream backward when both plan their K-splits alike (the side-stream
    "background" plans use fewer pieces per tile, which only moves fp32 summation order: checked to 1e-5 with the default
    plans), also across accumulation, back-to-back backwards and with the bucket callback."""
    cfg, sd, sd_bf, m = tiny
    ids, lab = (torch.from_numpy(golden_npz[k]) for k in ("pad_ids", "pad_labels"))

    def run(two, timed=0):
        m.engine.set_option("bwd_wgrad_stream", two)
        m.engine.set_option("time_families", timed)  # timing-event pairs around every launch must not change a bit
        m.zero_grad()
        got = []
        for rep in range(3):  # back-to-back backwards: the side stream of one must not run into the next
            m(input_ids=ids, labels=lab, return_logits=False)
            m.backward(0.5 if rep else 1.0, 1, lambda off, cnt, stream=None: got.append((off, cnt)))
        torch.cuda.synchronize()
        return m.flat_grads.clone(), got

    try:
        m.engine.set_option("gemm_tn_bal_bg_max_split", 8)  # same split plans on both paths
        m.engine.set_option("gemm_tn224_bg_min_m", 1 << 30)
        m.engine.set_option("gemm_nt224", 0)
        outs = [run(0), run(1), run(1), run(1, 1)]
        assert all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
        assert all(outs[0][1] == o[1] for o in outs[1:])
        fam = m.engine.family_ms()  # the last forward + backward, in launch order
        names = [n for n, _ in fam]
        L = cfg.n_layers
        assert names.count("wgu_wgrad") == L and names.count("gateup_fwd") == L and names.count("attn_bwd") == L
        assert names.count("norm_bwd") == 2 * L + 1 and all(ms >= 0.0 for _, ms in fam)
        m.engine.set_option("gemm_tn_bal_bg_max_split", 4)
        m.engine.set_option("gemm_tn224_bg_min_m", 4096)
        m.engine.set_option("gemm_nt224", 1)
        dflt = run(1)
        assert rel_err(dflt[0], outs[0][0]) <= 1e-5
    finally:
        m.engine.set_option("bwd_wgrad_stream", 1)
        m.engine.set_option("time_families", 0)
        m.engine.set_option("gemm_tn_bal_bg_max_split", 4)
        m.engine.set_option("gemm_tn224_bg_min_m", 4096)
        m.engine.set_option("gemm_nt224", 1)


def test_pack_unpack_grads_bf16_are_exact(tiny):
    """slam_pack_grads_bf16 / slam_unpack_grads_bf16 (the bf16 gradient exchange's staging passes, engine kernels on the
    communication stream): round-to-nearest-even packing bit-identical to a tensor conversion, widening exact, ranges
    outside [offset, offset + count) untouched."""
    cfg, sd, sd_bf, m = tiny
    n = m.engine.n_params
    g = torch.Generator(device="cuda").manual_seed(1)
    m.flat_grads.copy_(torch.randn(n, device="cuda", generator=g) * torch.logspace(-6, 3, n, device="cuda"))
    ref = m.flat_grads.clone()
    stage = torch.full((n,), 7.0, dtype=torch.bfloat16, device="cuda")
    off, cnt = 4096, ((n - 4096) // 8) * 4 + 4
    m.engine.pack_grads_bf16(off, cnt, stage[off:off + cnt])
    torch.cuda.synchronize()
    assert torch.equal(stage[off:off + cnt], ref[off:off + cnt].to(torch.bfloat16))
    assert bool((stage[:off] == 7).all()) and bool((stage[off + cnt:] == 7).all())
    m.engine.unpack_grads_bf16(off, cnt, stage[off:off + cnt])
    torch.cuda.synchronize()
    assert torch.equal(m.flat_grads[off:off + cnt], ref[off:off + cnt].to(torch.bfloat16).float())
    assert torch.equal(m.flat_grads[:off], ref[:off]) and torch.equal(m.flat_grads[off + cnt:], ref[off + cnt:])
    m.zero_grad()


def test_clip_and_adamw_step_vs_oracle(tiny, golden_npz):
    cfg, sd, sd_bf, m = tiny
    ids, am, lab = (torch.from_numpy(golden_npz[k]) for k in ("pad_ids", "pad_mask", "pad_labels"))
    m.load_state_dict(sd)
    m.zero_grad()
    m(input_ids=ids, labels=lab, return_logits=False)
    m.backward()
    grads = {k: v.detach().cpu().clone() for k, v in m.named_grads()}
    norm_out = torch.zeros(2, device=m.device)
    mbuf, vbuf = torch.zeros_like(m.flat_master), torch.zeros_like(m.flat_master)
    m.engine.grad_norm(0.5, norm_out)
    tot, coef = O.clip_coef(grads, 0.5)
    torch.cuda.synchronize()
    assert abs(float(norm_out[0]) - tot) <= 1e-4 * tot and abs(float(norm_out[1]) - coef) <= 1e-4 * coef
    p_ref = {k: v.clone() for k, v in sd.items()}
    m_ref = {k: torch.zeros_like(v) for k, v in sd.items()}
    v_ref = {k: torch.zeros_like(v) for k, v in sd.items()}
    for step in (1, 2, 3):
        m.engine.adamw_step(m.flat_master, mbuf, vbuf, norm_out, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, zero_grad=False)
        for k in sd:
            O.adamw_update(p_ref[k], grads[k] * coef, m_ref[k], v_ref[k], step, 1e-3, wd=0.01)
    torch.cuda.synchronize()
    new = m.state_dict(torch.float32)
    for k in sd:
        assert torch.allclose(new[k], p_ref[k], rtol=2e-5, atol=2e-6), k
        assert torch.equal(m.state_dict(torch.bfloat16)[k].float(), new[k].to(torch.bfloat16).float()), k
    m.load_state_dict(sd)


def _check_all_grads(m, grads_ref, tag, big=0.999, small=0.99):
    """Every gradient tensor against the oracle: cosine >= `big` for matrices, >= `small` for the bias / norm vectors
    (SURVEY.md §8c). Prints the worst of each class; returns them."""
    worst = {True: (2.0, ""), False: (2.0, "")}
    bad = []
    for k, gv in m.named_grads():
        is_small = k.endswith(".bias") or k.endswith("norm.weight")
        c = cosine(gv, grads_ref[k])
        if c < worst[is_small][0]:
            worst[is_small] = (c, k)
        if c < (small if is_small else big):
            bad.append((k, round(c, 5)))
    print(f"[parity] {tag}: worst matrix-gradient cosine {worst[False][0]:.5f} ({worst[False][1]}), "
          f"worst vector-gradient cosine {worst[True][0]:.5f} ({worst[True][1]})")
    assert not bad, bad
    return worst


def test_slam358m_loss_and_grads_vs_oracle():
    """Full-size Slam-358M (24 L, H 896, 14/2 heads, I 4864, V 502) at the full context length (B 1, T 1024) against the
    fp32 CPU oracle: loss, logits and EVERY one of the 290 gradient tensors."""
    B, T = 1, 1024
    cfg = O.SLAM_358M
    sd = O.init_weights(cfg, seed=0)
    sd_bf = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    m = _mk(cfg, sd, max_tokens=8192)
    assert m.num_parameters() == 358_347_904
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(2, 502, (B, T), generator=g)
    ids[:, 0] = 1
    loss_ref, logits_ref, grads_ref = O.forward_loss_grads(cfg, sd_bf, ids, ids)
    m.zero_grad()
    out = m(input_ids=ids, labels=ids)
    m.backward()
    torch.cuda.synchronize()
    print("slam358m loss engine/oracle", float(out.loss), float(loss_ref))
    assert abs(float(out.loss) - float(loss_ref)) <= 2e-2
    check("slam358m logits", out.logits.float().cpu(), logits_ref, 2e-2)
    assert len(list(m.named_grads())) == 290
    _check_all_grads(m, grads_ref, "Slam-358M T=1024")

    # ---- size-independent properties at the BASELINE.json shape (B=8, T=1024) ----------------
    ids8 = torch.randint(2, 502, (8, 1024), generator=g)
    ids8[:, 0] = 1
    m.zero_grad()
    o8 = m(input_ids=ids8, labels=ids8, return_logits=False)
    m.backward()
    l8 = float(o8.loss)
    assert math.isfinite(l8) and abs(l8 - math.log(502)) < 0.5
    # the BASELINE shape itself against the oracle (forward only: 8 x 1024 tokens of fp32 CPU work)
    with torch.no_grad():
        loss8_ref = float(O.compute_loss(O.model_forward(cfg, sd_bf, ids8), ids8))
    print("slam358m B=8 T=1024 loss engine/oracle", l8, loss8_ref)
    assert abs(l8 - loss8_ref) <= 2e-2
    assert bool(torch.isfinite(m.flat_grads).all())
    g8 = m.flat_grads.clone()
    # batch-row permutation leaves the token-mean loss and the summed gradient unchanged
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
    m.zero_grad()
    o8p = m(input_ids=ids8[perm], labels=ids8[perm], return_logits=False)
    m.backward()
    torch.cuda.synchronize()
    assert abs(float(o8p.loss) - l8) <= 1e-5
    assert rel_err(m.flat_grads, g8) <= 1e-3
    # the loss of the whole batch is the token-weighted mean of per-row losses (checksum of checksums)
    rows = [float(m(input_ids=ids8[i:i + 1], labels=ids8[i:i + 1], return_logits=False).loss) for i in range(8)]
    assert abs(sum(rows) / 8 - l8) <= 2e-3


# ---- config-4-shaped dims: head_dim 128, vocabulary beyond 512, rope_theta 1e6 ----------
def test_wide_config_vs_reference_golden_and_oracle(wide_golden):
    g, cfgd, seed, bias_std, jit = wide_golden
    cfg = O.OracleConfig(**cfgd)
    sd = O.init_weights(cfg, seed=seed, bias_std=bias_std, norm_jitter=jit)
    sd_bf = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    m = _mk(cfg, sd)
    ids, am, lab = (torch.from_numpy(g[k]) for k in ("pad_ids", "pad_mask", "pad_labels"))
    out = m(input_ids=ids, attention_mask=am, labels=lab)
    valid = am.bool()
    check("wide logits vs reference golden", out.logits.float().cpu()[valid], torch.from_numpy(g["pad_logits"])[valid], 2e-2)
    assert abs(float(out.loss) - float(g["pad_loss_mean"])) <= 2e-2
    loss_ref, logits_ref, grads_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, attention_mask=am)
    m.zero_grad()
    out = m(input_ids=ids, attention_mask=am, labels=lab)
    m.backward()
    torch.cuda.synchronize()
    assert abs(float(out.loss) - float(loss_ref)) <= 5e-3
    check("wide logits vs oracle", out.logits.float().cpu()[valid], logits_ref[valid], 1e-2)
    grads = dict(m.named_grads())
    for k, gv in grads.items():
        small = k.endswith(".bias") or k.endswith("norm.weight")
        c = cosine(gv, grads_ref[k])
        assert c >= (0.99 if small else 0.999), f"{k}: cosine {c:.5f}"
        assert abs(float(gv.norm()) / float(grads_ref[k].norm()) - 1) <= 3e-2, k
    rows = torch.from_numpy(g["pad_embed_grad_rows"])
    assert cosine(grads["lm.model.embed_tokens.weight"].cpu()[rows], torch.from_numpy(g["pad_embed_grad"])) >= 0.99
    # packed batch (varlen attention, per-segment RoPE restart) and forward-only likelihood
    ids, pos, lab = (torch.from_numpy(g[k]) for k in ("pack_ids", "pack_pos", "pack_labels"))
    outp = m(input_ids=ids, position_ids=pos, labels=lab)
    check("wide packed logits vs reference", outp.logits.float().cpu(), torch.from_numpy(g["pack_logits"]), 2e-2)
    assert abs(float(outp.loss) - float(g["pack_loss_mean"])) <= 2e-2
    _, _, gp_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, position_ids=pos, packed=True)
    m.zero_grad()
    m(input_ids=ids, position_ids=pos, labels=lab, return_logits=False)
    m.backward()
    for k, gv in m.named_grads():
        small = k.endswith(".bias") or k.endswith("norm.weight")
        assert cosine(gv, gp_ref[k]) >= (0.99 if small else 0.999), k
    ll = m.log_likelihood(torch.from_numpy(g["pad_ids"]), True).cpu().numpy()
    assert np.allclose(ll, g["ll_mean"], rtol=5e-3, atol=2e-2)
    # modality-restricted scoring: ignore_tokens are -inf inside the CE kernel (large-vocabulary kernel here)
    lli = m.log_likelihood(torch.from_numpy(g["pad_ids"]), False, ignore_tokens=g["ll_ignore_tokens"].tolist()).cpu().numpy()
    assert np.allclose(lli, g["ll_ignore_sum"], rtol=5e-3, atol=0.5), (lli, g["ll_ignore_sum"])
    ll2 = m.log_likelihood(torch.from_numpy(g["pad_ids"]), True).cpu().numpy()
    assert np.array_equal(ll, ll2)  # the mask is reset afterwards
    # determinism of the large-vocabulary path (token-ordered embedding scatter, no float atomics)
    g1 = m.flat_grads.clone()
    m.zero_grad()
    m(input_ids=ids, position_ids=pos, labels=lab, return_logits=False)
    m.backward()
    assert torch.equal(m.flat_grads, g1)


def test_qwen15b_shaped_layers_vs_oracle():
    """Two layers of the Qwen2.5-1.5B shape (H 1536, 12/2 heads of 128, I 8960) with a 20k vocabulary, T 256."""
    cfg = O.OracleConfig(n_layers=2, hidden=1536, n_heads=12, n_kv_heads=2, head_dim=128, intermediate=8960, vocab=20003,
                         rope_theta=1000000.0)
    sd = O.init_weights(cfg, seed=1)
    sd_bf = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    m = _mk(cfg, sd, max_tokens=2048)
    gen = torch.Generator().manual_seed(99)
    ids = torch.randint(2, cfg.vocab, (2, 256), generator=gen)
    ids[:, 0] = 1
    ids[1, 200:] = 0
    lab = ids.clone()
    lab[ids == 0] = -100
    am = (ids != 0).long()
    loss_ref, logits_ref, grads_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, attention_mask=am)
    m.zero_grad()
    out = m(input_ids=ids, attention_mask=am, labels=lab)
    m.backward()
    torch.cuda.synchronize()
    print("qwen1.5b-shaped loss engine/oracle", float(out.loss), float(loss_ref))
    assert abs(float(out.loss) - float(loss_ref)) <= 2e-2
    check("qwen1.5b-shaped logits", out.logits.float().cpu()[am.bool()], logits_ref[am.bool()], 2e-2)
    for k, gv in m.named_grads():
        small = k.endswith(".bias") or k.endswith("norm.weight")
        c = cosine(gv, grads_ref[k])
        assert c >= (0.99 if small else 0.998), f"{k}: cosine {c:.5f}"


def test_full_vocab_152k_properties():
    """V = 152,167 (Qwen2.5 tokenizer + 500 units + 2 markers): size-independent properties only."""
    cfg = O.OracleConfig(n_layers=1, hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=1024, vocab=152167,
                         rope_theta=1000000.0)
    m = _mk(cfg, None, max_tokens=4096)
    gen = torch.Generator().manual_seed(7)
    ids = torch.randint(2, cfg.vocab, (2, 1024), generator=gen)
    ids[:, 0] = 1
    out = m(input_ids=ids, labels=ids, return_logits=False)
    l0 = float(out.loss)
    assert abs(l0 - math.log(cfg.vocab)) < 0.5, l0
    m.zero_grad()
    m.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.flat_grads).all())
    g1 = m.flat_grads.clone()
    eg = dict(m.named_grads())["lm.model.embed_tokens.weight"]
    assert eg.shape == (cfg.vocab, cfg.hidden)
    # rows of ids that occur get a gather-side contribution far above the tied-head one
    seen = torch.zeros(cfg.vocab, dtype=torch.bool)
    seen[ids.flatten()] = True
    rn = eg.float().norm(dim=1).cpu()
    assert float(rn[seen].mean()) > 3 * float(rn[~seen].mean())
    perm = torch.tensor([1, 0])
    m.zero_grad()
    o2 = m(input_ids=ids[perm], labels=ids[perm], return_logits=False)
    m.backward()
    torch.cuda.synchronize()
    assert abs(float(o2.loss) - l0) <= 1e-5
    assert rel_err(m.flat_grads, g1) <= 2e-3


# ---- BASELINE.json configs[3] at its workload: 28 layers, V = 152,167, packed rows, ctx 2048 ----------------------------
def _packed_row(lens, vocab, unit_lo, gen):
    """DataCollatorWithFlattening layout ([1, sum T]): 45 % unit ids / 55 % text ids per sequence,
    position_ids restarting per sequence, labels -100 at sequence starts."""
    ids, pos, lab = [], [], []
    for n in lens:
        unit = torch.rand(n, generator=gen) < 0.45
        t = torch.where(unit, torch.randint(unit_lo, vocab, (n,), generator=gen), torch.randint(2, unit_lo, (n,), generator=gen))
        t[0] = 1
        l = t.clone()
        l[0] = -100
        ids.append(t); pos.append(torch.arange(n)); lab.append(l)
    cat = lambda xs: torch.cat(xs)[None]  # noqa: E731
    return cat(ids), cat(pos), cat(lab)


def test_configs3_full_depth_packed_vs_oracle():
    """The interleaved speech-text model at full depth and full vocabulary (Qwen2.5-1.5B body: 28 L, H 1536, 12/2 heads of
    128, I 8960, rope_theta 1e6; V = 152,167) on ONE packed row of 2048 tokens in four segments (the longest spans
    1100 tokens, i.e. beyond any single 1024 window) against the fp32 CPU oracle: loss <= 2e-2, logits rel-RMS <= 2e-2,
    gradient cosine on ALL 338 tensors. Then the 16,384-token packed micro-batch of the bench workload through the
    size-independent properties: bit-identical repeat, and invariance under a permutation of the packed segments."""
    cfg = O.OracleConfig(n_layers=28, hidden=1536, n_heads=12, n_kv_heads=2, head_dim=128, intermediate=8960, vocab=152167,
                         rope_theta=1000000.0)
    import psutil
    wg = torch.Generator().manual_seed(2)  # (plain seeded normals: the hash-based golden initialiser takes 100 s at 1.5 B parameters)
    sd_bf = {}
    for k, shp in O.hf_keys(cfg):
        w = (1.0 + 0.1 * torch.randn(shp, generator=wg)) if k.endswith("norm.weight") else 0.02 * torch.randn(shp, generator=wg)
        sd_bf[k] = w.to(torch.bfloat16).float()
    sd_bf["lm.model.embed_tokens.weight"][0].zero_()
    m = _mk(cfg, sd_bf, max_tokens=16384)
    gen = torch.Generator().manual_seed(99)
    # the fp32 autograd oracle keeps ~14 MB per token of activations over 28 layers: 2048 tokens need ~36 GB of host memory
    avail = psutil.virtual_memory().available / 2 ** 30
    lens = [1100, 500, 64, 384] if avail >= 48 else [600, 300, 64, 60]
    print(f"host memory available {avail:.0f} GiB -> packed row of {sum(lens)} tokens")
    ids, pos, lab = _packed_row(lens, cfg.vocab, 151667, gen)
    assert ids.shape == (1, sum(lens))
    loss_ref, logits_ref, grads_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, position_ids=pos, packed=True)
    # Logit tolerance at this depth and width: the 2e-2 of SURVEY.md §8c was calibrated on shallow models. The reference's
    # OWN precision (bf16 tensors between modules, oracle `bf16_acts`, pinned on the tiny model against the reference's
    # bf16-autocast run in tests/test_oracle_golden.py) sits 2.9e-2 away from the fp32 run here (1.8e-2 for Slam-358M): the
    # engine has to be at least as close to fp32 as that path is (+10 % slack), and never needs to beat 2e-2.
    # Measured HERE, on this batch, on every run (~30 s of host time; round 4 kept a recorded 2.999e-2 instead).
    with torch.no_grad():
        emu = O.model_forward(cfg, sd_bf, ids, position_ids=pos, packed=True, bf16_acts=True)
    emu_dev = rel_err(emu, logits_ref)
    del emu
    del sd_bf
    m.zero_grad()
    out = m(input_ids=ids, position_ids=pos, labels=lab)
    m.backward()
    torch.cuda.synchronize()
    print("configs[3] full-depth loss engine/oracle", float(out.loss), float(loss_ref))
    assert abs(float(out.loss) - float(loss_ref)) <= 2e-2
    tol = max(2e-2, 1.1 * emu_dev)
    print(f"[parity] configs[3] bf16-path emulation vs fp32: logits rel-rms {emu_dev:.3e} -> engine tolerance {tol:.3e}")
    check("configs[3] logits", out.logits.float().cpu(), logits_ref, tol)
    del logits_ref
    assert len(list(m.named_grads())) == 338
    # Gradient bar at this depth: the bf16-path emulation of the oracle (the reference's own precision) reaches 0.99874 on
    # its worst matrix (k_proj of layer 26; q / k projections of the top layers sit at 0.9987-0.9988) and 0.99862 on its
    # worst vector against the fp32 run on this very batch - the 0.999 of SURVEY.md §8c is not attainable by ANY bf16
    # implementation here. Stated bar: >= 0.998 matrices (engine measured 0.99898 worst), >= 0.99 vectors.
    _check_all_grads(m, grads_ref, "configs[3] 28 L, V 152167, packed 2048", big=0.998)
    del grads_ref

    # 16,384 packed tokens (the bench's micro-batch): determinism and segment-permutation invariance
    lens16 = [2048, 1531, 64, 2000, 777, 2048, 1200, 300, 1900, 2048, 2468 - 2000]
    lens16.append(16384 - sum(lens16))
    assert sum(lens16) == 16384 and min(lens16) > 0
    seqs = [_packed_row([n], cfg.vocab, 151667, gen) for n in lens16]

    def run(order):
        i_, p_, l_ = (torch.cat([seqs[j][k] for j in order], 1) for k in range(3))
        m.zero_grad()
        o = m(input_ids=i_, position_ids=p_, labels=l_, return_logits=False)
        m.backward()
        torch.cuda.synchronize()
        return float(o.loss), m.flat_grads.clone()

    l0, g0 = run(range(len(seqs)))
    l1, g1 = run(range(len(seqs)))
    assert math.isfinite(l0) and abs(l0 - math.log(cfg.vocab)) < 0.5
    assert l0 == l1 and torch.equal(g0, g1)  # no atomics anywhere: the same bits every run
    l2, g2 = run([3, 0, 11, 7, 1, 9, 2, 10, 5, 4, 8, 6])
    # A permutation of the packed segments moves every sequence against the 64 / 128-row attention tiles: the forward O
    # (bf16) changes in its last bit, D = rowsum(dO * O) with it, and dS = P (dP - D) - a difference of nearly equal terms -
    # by ~1 % per layer on the q / k projections (tools/diag_perm.py: 1.2 % worst tensor at 2 layers, 3.7 % of the flat
    # gradient at 28). That is the precision's own noise (the same tensors sit at cosine 0.999 against the fp32 oracle): the
    # permuted run has to agree to that bar; the loss (ln 152167 = 11.9) to 5e-4: the attention probabilities are rounded
    # to bf16 relative to a running max that depends on where a sequence falls against the 64-key tiles, so O - and with it
    # the loss - moves by bf16 rounding noise under the permutation (measured 1.7e-4 = 1.4e-5 relative at 28 layers).
    assert abs(l2 - l0) <= 5e-4 and rel_err(g2, g0) <= 6e-2 and cosine(g2, g0) >= 0.998


def test_degenerate_batches_vs_oracle(tiny):
    """Edge batches a data-parallel rank can meet (unit_lm.py:13-29 semantics, checked against the oracle):
      * one row whose labels are ALL -100 beside a normal row (mean reduction and the num_items_in_batch form);
      * a micro-batch with NO valid target at all under the num_items_in_batch form (the step's tokens sit on the other ranks):
        loss exactly 0, every gradient exactly 0, nothing non-finite - the reference's `sum / num_items` gives the same;
      * the shortest shapes: B = 1 with T = 2 (one target) and T = 1 (no target after the shift);
      * a row of 1 real token followed by padding (attention over a single key)."""
    cfg, sd, sd_bf, m = tiny
    g = torch.Generator().manual_seed(5)

    def run(ids, lab, am=None, n=None):
        m.zero_grad()
        out = m(input_ids=ids, attention_mask=am, labels=lab, **({"num_items_in_batch": n} if n is not None else {}))
        m.backward()
        torch.cuda.synchronize()
        return out, {k: v.clone() for k, v in m.named_grads()}

    ids = torch.randint(2, cfg.vocab, (2, 40), generator=g)
    ids[:, 0] = 1
    lab = ids.clone()
    lab[1, :] = -100
    for n in (None, 39):
        out, grads = run(ids, lab, n=n)
        l_ref, lg_ref, g_ref = O.forward_loss_grads(cfg, sd_bf, ids, lab, **({"num_items_in_batch": float(n)} if n else {}))
        assert abs(float(out.loss) - float(l_ref)) <= 2e-2, (n, float(out.loss), float(l_ref))
        check(f"one fully ignored row, num_items={n}: logits", out.logits.float().cpu(), lg_ref, 2e-2)
        k = "lm.model.layers.0.mlp.down_proj.weight"
        assert cosine(grads[k].cpu(), g_ref[k]) >= 0.999
    # no valid target on this rank, the global count comes from elsewhere
    lab0 = torch.full_like(ids, -100)
    out, grads = run(ids, lab0, n=100)
    assert float(out.loss) == 0.0
    assert torch.isfinite(out.logits.float()).all()
    for k, v in grads.items():
        assert torch.isfinite(v).all() and float(v.abs().max()) == 0.0, k
    # shortest shapes
    for T in (2, 1):
        i2 = torch.tensor([[1, 7][:T]])
        out, grads = run(i2, i2.clone(), n=1)
        l_ref, lg_ref, g_ref = O.forward_loss_grads(cfg, sd_bf, i2, i2.clone(), num_items_in_batch=1.0)
        assert abs(float(out.loss) - float(l_ref)) <= 2e-2, (T, float(out.loss), float(l_ref))
        check(f"B=1 T={T}: logits", out.logits.float().cpu(), lg_ref, 2e-2)
        assert all(torch.isfinite(v).all() for v in grads.values())
    # one real token, then padding
    i3 = torch.tensor([[1, 0, 0, 0, 0, 0], [1, 9, 11, 13, 0, 0]])
    am = torch.tensor([[1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0]])
    l3 = torch.where(am.bool(), i3, torch.full_like(i3, -100))
    out, grads = run(i3, l3, am=am)
    l_ref, lg_ref, g_ref = O.forward_loss_grads(cfg, sd_bf, i3, l3, attention_mask=am)
    assert abs(float(out.loss) - float(l_ref)) <= 2e-2
    check("single-token row: logits at valid positions", out.logits.float().cpu()[am.bool()], lg_ref[am.bool()], 2e-2)
    assert all(torch.isfinite(v).all() for v in grads.values())
