"""CPU tests: the C oracle against the golden vectors (generated from the reference's own PyTorch
oracle, tests/golden/make_golden.py) and against torch SDPA (the fp32 plumbing configuration)."""
import numpy as np
import pytest

from oracle import fa_oracle
from util import bits_to_f32, expected_lse, golden_names, load_golden


def _lens(meta, key, full):
    v = meta.get(key)
    return list(v) if v else [full] * meta["b"]


def run_oracle_dense(g, dtype=None, **kw):
    m = g["meta"]
    dt = dtype or m["dtype"]
    q, k, v = (bits_to_f32(g[x], m["dtype"]) for x in "qkv")
    args = dict(dtype=dt, causal=m["causal"], window=tuple(m["window"]),
                alibi_slopes=g.get("alibi_slopes"))
    args.update(kw)
    if m.get("qlens"):
        qlens, klens = m["qlens"], m["klens"]
        cq = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
        ck = np.concatenate([[0], np.cumsum(klens)]).astype(np.int32)
        qu = np.concatenate([q[i, :n] for i, n in enumerate(qlens)])
        ku = np.concatenate([k[i, :n] for i, n in enumerate(klens)])
        vu = np.concatenate([v[i, :n] for i, n in enumerate(klens)])
        out_u, lse = fa_oracle.attention_fwd(qu, ku, vu, cu_seqlens_q=cq, cu_seqlens_k=ck,
                                             max_seqlen_q=max(qlens), max_seqlen_k=max(klens), **args)
        out = np.zeros_like(q)
        for i, n in enumerate(qlens):
            out[i, :n] = out_u[cq[i]:cq[i + 1]]
        return out, lse
    return fa_oracle.attention_fwd(q, k, v, **args)


def check_out(g, out, lse, mult=2.0):
    m = g["meta"]
    ref = g["out_ref"]
    qlens = _lens(m, "qlens", m["sq"])
    tol = mult * float(g["err_out_pt"]) + 1e-5
    for i, n in enumerate(qlens):
        err = np.abs(out[i, :n] - ref[i, :n]).max()
        assert err <= tol, (m["name"], i, err, tol)
    # LSE: fp32 quantity, compare to the textbook logsumexp (+ causal-ALiBi shift)
    lse_ref = expected_lse(g)
    for i, n in enumerate(qlens):
        a, r = lse[i, :, :n], lse_ref[i, :, :n]
        fin = np.isfinite(r)
        assert np.array_equal(np.isposinf(a), np.isposinf(r)), m["name"]
        assert np.abs(a[fin] - r[fin]).max() <= 2e-3 * max(1.0, np.abs(r[fin]).max()), m["name"]


@pytest.mark.parametrize("name", golden_names("dense"))
def test_oracle_fwd_vs_golden(name):
    g = load_golden(name)
    out, lse = run_oracle_dense(g)
    check_out(g, out, lse)


@pytest.mark.parametrize("name", golden_names("dense"))
def test_oracle_fwd_fp32_mode_is_tight(name):
    """With rounding disabled the oracle must match the fp32 reference to fp32 accuracy: pins the
    masks, ALiBi forms, GQA map, varlen offsets and online-softmax algebra independent of 16-bit
    noise."""
    g = load_golden(name)
    out, _ = run_oracle_dense(g, dtype="fp32")
    m = g["meta"]
    for i, n in enumerate(_lens(m, "qlens", m["sq"])):
        scale = max(1.0, np.abs(g["out_ref"][i, :n]).max())
        assert np.abs(out[i, :n] - g["out_ref"][i, :n]).max() <= 3e-5 * scale, name


@pytest.mark.parametrize("name", [n for n in golden_names("dense") if "varlen" not in n][:8])
def test_oracle_split_kv_matches_standard(name):
    g = load_golden(name)
    out1, lse1 = run_oracle_dense(g, dtype="fp32")
    out2, lse2 = run_oracle_dense(g, dtype="fp32", num_splits=3, block_n=32)
    assert np.abs(out1 - out2).max() <= 1e-5 * max(1.0, np.abs(out1).max())
    fin = np.isfinite(lse1)
    assert np.array_equal(np.isposinf(lse1), np.isposinf(lse2))
    assert np.abs(lse1[fin] - lse2[fin]).max() <= 1e-4


@pytest.mark.parametrize("name", [n for n in golden_names("dense")])
def test_oracle_bwd_vs_golden(name):
    g = load_golden(name)
    if "dq_ref" not in g:
        pytest.skip("fixture has no gradients")
    m = g["meta"]
    q, k, v, do = (bits_to_f32(g[x], m["dtype"]) for x in "qkvg")
    args = dict(dtype=m["dtype"], causal=m["causal"], window=tuple(m["window"]),
                alibi_slopes=g.get("alibi_slopes"))
    if m.get("qlens"):
        qlens, klens = m["qlens"], m["klens"]
        cq = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
        ck = np.concatenate([[0], np.cumsum(klens)]).astype(np.int32)
        cat = lambda x, ls: np.concatenate([x[i, :n] for i, n in enumerate(ls)])
        qu, dou, ku, vu = cat(q, qlens), cat(do, qlens), cat(k, klens), cat(v, klens)
        vargs = dict(cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(qlens), max_seqlen_k=max(klens))
        out_u, lse = fa_oracle.attention_fwd(qu, ku, vu, **args, **vargs)
        dq_u, dk_u, dv_u, _ = fa_oracle.attention_bwd(dou, qu, ku, vu, out_u, lse, **args, **vargs)
        dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
        for i in range(m["b"]):
            dq[i, :qlens[i]] = dq_u[cq[i]:cq[i + 1]]
            dk[i, :klens[i]] = dk_u[ck[i]:ck[i + 1]]
            dv[i, :klens[i]] = dv_u[ck[i]:ck[i + 1]]
    else:
        out, lse = fa_oracle.attention_fwd(q, k, v, **args)
        dq, dk, dv, _ = fa_oracle.attention_bwd(do, q, k, v, out, lse, **args)
    mult = 3.0 if m.get("qlens") else 2.0  # tests/test_flash_attn.py:1096-1098, :1402-1404
    add = 1e-4
    if "_x5_" in name or "_x3_" in name:       # overflow inputs: 5x + 1e-3 (:2220-2228)
        mult, add = 5.0, 1e-3
    for name_, a in (("dq", dq), ("dk", dk), ("dv", dv)):
        ref = g[name_ + "_ref"]
        if m.get("qlens"):  # padded positions of the reference are zero-gradient
            lens = m["qlens"] if name_ == "dq" else m["klens"]
            for i, n in enumerate(lens):
                ref = ref.copy()
                ref[i, n:] = 0
        err = np.abs(a - ref).max()
        tol = mult * float(g["err_" + name_ + "_pt"]) + add
        assert err <= tol, (name, name_, err, tol)


@pytest.mark.parametrize("name", golden_names("kvcache"))
def test_oracle_kvcache_vs_golden(name):
    g = load_golden(name)
    m = g["meta"]
    q = bits_to_f32(g["q"], m["dtype"])
    args = dict(dtype=m["dtype"], causal=m["causal"], window=tuple(m["window"]),
                alibi_slopes=g.get("alibi_slopes"), cache_seqlens=g["cache_seqlens"])
    if m["page"]:
        k, v = bits_to_f32(g["k_paged"], m["dtype"]), bits_to_f32(g["v_paged"], m["dtype"])
        args["block_table"] = g["block_table"]
    else:
        k, v = bits_to_f32(g["k"], m["dtype"]), bits_to_f32(g["v"], m["dtype"])
    mult = 5.0 if m["alibi"] else 3.0  # tests/test_flash_attn.py:2075-2076
    for ns in (1, 4):
        out, lse = fa_oracle.attention_fwd(q, k, v, num_splits=ns, **args)
        err = np.abs(out - g["out_ref"]).max()
        assert err <= mult * float(g["err_out_pt"]) + 1e-5, (name, ns, err)


def test_config1_fp32_vs_torch_sdpa():
    """Plumbing configuration: batch=2 seqlen=128 nheads=4 headdim=64 fp32 non-causal vs torch SDPA."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 128, 4, 64) for _ in range(3))
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    out, _ = fa_oracle.attention_fwd(q.numpy(), k.numpy(), v.numpy(), dtype="fp32")
    assert np.abs(out - ref.transpose(1, 2).numpy()).max() <= 1e-5


def test_rounding_matches_numpy():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4096).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 300, 7e4)])
    f16 = np.array([fa_oracle.lib().fa_oracle_round(float(t), 1) for t in x], np.float32)
    with np.errstate(over="ignore"):
        assert np.array_equal(f16, x.astype(np.float16).astype(np.float32))
    bf = np.array([fa_oracle.lib().fa_oracle_round(float(t), 2) for t in x], np.float32)
    assert np.array_equal(bf, fa_oracle.round_to(x, "bf16"))
    import torch
    assert np.array_equal(bf, torch.from_numpy(x).to(torch.bfloat16).float().numpy())


def test_fold_window_rules():
    # flash_api.cpp:289-294, :121-124
    assert fa_oracle.fold_window(128, 128, False, True, (-1, -1)) == (True, 128, 0, False) or \
        fa_oracle.fold_window(128, 128, False, True, (-1, -1))[0] is True
    c, wl, wr, loc = fa_oracle.fold_window(1, 64, False, True, (-1, -1))
    assert (c, loc) == (False, False) and wl < 0 and wr < 0      # sq == 1 drops causal
    c, wl, wr, loc = fa_oracle.fold_window(1, 64, True, True, (-1, -1))
    assert c is True                                              # ... unless ALiBi
    c, wl, wr, loc = fa_oracle.fold_window(64, 64, False, False, (100, 3))
    assert (c, wl, wr, loc) == (False, 64, 3, True)               # wl >= sk -> -1 -> filled with sk
    c, wl, wr, loc = fa_oracle.fold_window(64, 64, False, False, (5, -1))
    assert (c, wl, wr, loc) == (False, 5, 64, True)


def test_num_splits_heuristic():
    # flash_api.cpp:152-157 worked example: 48 blocks on 108 SMs -> 2 splits
    assert fa_oracle.num_splits_heuristic(48, 108, 64, 128) == 2
    assert fa_oracle.num_splits_heuristic(512, 512, 64, 128) == 1
