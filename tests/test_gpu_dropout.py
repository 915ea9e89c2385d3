"""GPU parity tests of dropout and return_attn_probs (csrc/flash_attn/src/dropout.h, philox.cuh;
compiled out of the reference fork, setup.py:204, but part of its Python signatures).

The keep mask must be the reference kernel's Philox pattern bit for bit (checked through S_dmask's
sign plane against the oracle's restatement), outputs and gradients follow the reference tests'
rules against attention_ref(dropout_p, dropout_mask) fixtures (2x, tests/test_flash_attn.py:1087-1098)
and the C oracle to a few ulp."""
import numpy as np
import pytest
import torch

from util import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _inputs(g):
    from gpu_util import to_dev
    m = g["meta"]
    q, k, v, do = (to_dev(g[x], m["dtype"]) for x in "qkvg")
    slopes = torch.from_numpy(g["alibi_slopes"]).cuda() if "alibi_slopes" in g else None
    return m, q, k, v, do, slopes


def _keep(g):
    m = g["meta"]
    n = m["b"] * m["h"] * m["sq"] * m["sk"]
    return np.unpackbits(g["keep_mask"])[:n].reshape(m["b"], m["h"], m["sq"], m["sk"]).astype(bool)


def _check_sdmask(sdm, keep, o_sdm, dtype):
    """Sign plane = dropped entries (wherever the magnitude is non-zero); magnitudes = the oracle's
    to one unit in the last place of the 16-bit type."""
    from gpu_util import f32
    b, h, sq, sk = keep.shape
    s = f32(sdm)[:, :, :sq, :sk]
    ref = o_sdm[:, :, :sq, :sk]
    nz = (s != 0) & (ref != 0)
    assert np.array_equal(np.signbit(s)[nz], ~keep[nz])
    eps = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    assert np.abs(np.abs(s) - np.abs(ref)).max() <= 2 * eps * max(1.0, np.abs(ref).max())
    assert (nz.sum() > 0.5 * (ref != 0).sum())


@pytest.mark.parametrize("name", golden_names("dropout"))
def test_dropout_cabi_vs_golden_and_oracle(name):
    import cabi
    from gpu_util import as_f32, f32, ulp_tol
    from oracle import fa_oracle
    g = load_golden(name)
    m, q, k, v, do, slopes = _inputs(g)
    keep = _keep(g)
    kw = dict(causal=m["causal"], window=tuple(m["window"]), alibi_slopes=slopes,
              p_dropout=m["p_dropout"], philox_seed=m["philox_seed"], philox_offset=m["philox_offset"])
    rng = torch.zeros(2, dtype=torch.int64, device="cuda")
    out, lse, sdm = cabi.fwd(q, k, v, return_softmax=True, rng_state=rng, **kw)
    assert [int(x) for x in rng.cpu()] == [m["philox_seed"], m["philox_offset"]]
    okw = dict(dtype=m["dtype"], causal=m["causal"], window=tuple(m["window"]),
               alibi_slopes=g.get("alibi_slopes"), p_dropout=m["p_dropout"],
               philox_seed=m["philox_seed"], philox_offset=m["philox_offset"])
    qf, kf, vf, dof = (as_f32(g[x], m["dtype"]) for x in "qkvg")
    o_out, o_lse, o_sdm = fa_oracle.attention_fwd(qf, kf, vf, return_softmax=True, block_n=64, **okw)
    _check_sdmask(sdm, keep, o_sdm, m["dtype"])
    assert np.abs(f32(out) - g["out_ref"]).max() <= 2 * float(g["err_out_pt"]) + 1e-5
    assert np.abs(f32(out) - o_out).max() <= ulp_tol(m["dtype"], o_out)
    fin = np.isfinite(o_lse)
    assert np.abs(f32(lse)[fin] - o_lse[fin]).max() <= 1e-3
    # the same result without the S_dmask output
    out2, _ = cabi.fwd(q, k, v, **kw)
    assert torch.equal(out, out2)
    # backward: Philox state by value and read back from the forward's rng_state
    dq, dk, dv, sd = cabi.bwd(do, q, k, v, out, lse, **kw)
    kw2 = dict(kw, philox_seed=0, philox_offset=0)
    dq2, dk2, dv2, _ = cabi.bwd(do, q, k, v, out, lse, rng_state=rng, **kw2)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    o_dq, o_dk, o_dv, o_sd = fa_oracle.attention_bwd(dof, qf, kf, vf, f32(out), f32(lse), **okw)
    for nm, a, o in (("dq", dq, o_dq), ("dk", dk, o_dk), ("dv", dv, o_dv)):
        err = np.abs(f32(a) - g[nm + "_ref"]).max()
        assert err <= 2 * float(g["err_" + nm + "_pt"]) + 1e-5, (name, nm, err)
        assert np.abs(f32(a) - o).max() <= 2 * ulp_tol(m["dtype"], o), (name, nm)
    assert np.abs(f32(sd)[:, :, :m["sq"]] - o_sd).max() <= 1e-3 * max(1.0, np.abs(o_sd).max())


@pytest.mark.parametrize("d,dtype", [(64, "bf16"), (128, "fp16"), (256, "bf16"), (96, "fp16")])
@pytest.mark.parametrize("causal", [False, True])
def test_dropout_python_api(d, dtype, causal):
    """flash_attn_func(dropout_p, return_attn_probs=True): rng_state comes from torch's generator,
    S_dmask's signs are the oracle mask for that state, forward / autograd backward match the
    oracle, and the generator's offset advances by b * h * 32 per call (flash_api.cpp:382)."""
    import vllm_flash_attn as fa
    from vllm_flash_attn.flash_attn_interface import _flash_attn_forward
    from gpu_util import TDT, f32, ulp_tol
    from oracle import fa_oracle
    torch.manual_seed(1234)
    b, sq, sk, h, hk, p = 2, 170, 333, 4, 2, 0.17
    q, k, v = (torch.randn(b, s, n, d, device="cuda", dtype=TDT[dtype]).requires_grad_()
               for s, n in ((sq, h), (sk, hk), (sk, hk)))
    do = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dtype])
    # the Philox state this call will draw
    st = _flash_attn_forward(q.detach(), k.detach(), v.detach(), p, d ** -0.5, causal, (-1, -1), None, False)[7]
    seed0, off0 = (int(x) for x in st.cpu())
    out, lse, sdm = fa.flash_attn_func(q, k, v, dropout_p=p, causal=causal, return_attn_probs=True)
    seed, off = seed0, off0 + b * h * 32
    keep = fa_oracle.dropout_mask(seed, off, b, h, sq, sk, p)
    okw = dict(dtype=dtype, causal=causal, p_dropout=p, philox_seed=seed, philox_offset=off)
    qf, kf, vf = (f32(x.detach()) for x in (q, k, v))
    o_out, o_lse, o_sdm = fa_oracle.attention_fwd(qf, kf, vf, return_softmax=True, block_n=64, **okw)
    _check_sdmask(sdm, keep, o_sdm, dtype)
    assert np.abs(f32(out.detach()) - o_out).max() <= ulp_tol(dtype, o_out)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    o_dq, o_dk, o_dv, _ = fa_oracle.attention_bwd(f32(do), qf, kf, vf, f32(out.detach()), f32(lse), **okw)
    for nm, a, o in (("dq", dq, o_dq), ("dk", dk, o_dk), ("dv", dv, o_dv)):
        assert np.abs(f32(a) - o).max() <= 2 * ulp_tol(dtype, o), nm
    # dropout fraction (tests/test_flash_attn.py:1085): keep probability (floor(0.83 * 255) + 1) / 256
    frac = 1.0 - keep.mean()
    assert abs(frac - (1 - (np.floor((1 - p) * 255) + 1) / 256)) < 0.01


def test_dropout_varlen_and_packed():
    """Ragged batch with dropout through flash_attn_varlen_qkvpacked_func: rows index the Philox
    pattern per sequence (the reference's m_block is relative to the sequence start)."""
    import vllm_flash_attn as fa
    from gpu_util import f32, ulp_tol
    from oracle import fa_oracle
    torch.manual_seed(7)
    lens, h, d, p = [76, 34, 146], 2, 64, 0.3
    total = sum(lens)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    qkv = torch.randn(total, 3, h, d, device="cuda", dtype=torch.float16).requires_grad_()
    do = torch.randn(total, h, d, device="cuda", dtype=torch.float16)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    seed, off = gen.initial_seed(), gen.get_offset()  # the Philox state the call will draw
    out, lse, sdm = fa.flash_attn_varlen_qkvpacked_func(qkv, cu, max(lens), dropout_p=p, causal=True,
                                                        return_attn_probs=True)
    assert gen.get_offset() == off + len(lens) * h * 32  # flash_api.cpp:382
    (dqkv,) = torch.autograd.grad(out, qkv, do)
    x = f32(qkv.detach())
    okw = dict(dtype="fp16", causal=True, p_dropout=p, philox_seed=seed, philox_offset=off,
               cu_seqlens_q=cu.cpu().numpy(), cu_seqlens_k=cu.cpu().numpy(), max_seqlen_q=max(lens),
               max_seqlen_k=max(lens))
    o_out, o_lse, o_sdm = fa_oracle.attention_fwd(x[:, 0], x[:, 1], x[:, 2], return_softmax=True, block_n=64, **okw)
    assert np.abs(f32(out.detach()) - o_out).max() <= ulp_tol("fp16", o_out)
    s = f32(sdm)
    for i, n in enumerate(lens):
        a, r = s[i, :, :n, :n], o_sdm[i, :, :n, :n]
        nz = (a != 0) & (r != 0)
        assert np.array_equal(np.signbit(a)[nz], np.signbit(r)[nz]) and nz.sum() > 0
    o_dq, o_dk, o_dv, _ = fa_oracle.attention_bwd(f32(do), x[:, 0], x[:, 1], x[:, 2], f32(out.detach()),
                                                  f32(lse), **okw)
    g = f32(dqkv)
    for i, o in enumerate((o_dq, o_dk, o_dv)):
        assert np.abs(g[:, i] - o).max() <= 2 * ulp_tol("fp16", o), i


def test_dropout_rejections():
    import vllm_flash_attn as fa
    q = torch.randn(1, 64, 2, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="p_dropout must be in"):
        fa.flash_attn_func(q, q, q, dropout_p=1.0)
    # without dropout return_attn_probs yields an EMPTY S_dmask tensor, like the reference
    # (flash_api.cpp:355-357 returns torch::empty({0}) in that slot)
    out, lse, sdm = fa.flash_attn_func(q, q, q, return_attn_probs=True)
    assert isinstance(sdm, torch.Tensor) and sdm.numel() == 0
    # repeatable for a fixed generator state
    torch.manual_seed(3)
    a = fa.flash_attn_func(q, q, q, dropout_p=0.4)
    torch.manual_seed(3)
    b = fa.flash_attn_func(q, q, q, dropout_p=0.4)
    assert torch.equal(a, b)
    c = fa.flash_attn_func(q, q, q, dropout_p=0.4)
    assert not torch.equal(a, c)


def test_dropout_forward_backward_under_graph_capture():
    """dropout_p > 0 inside a HIP graph: the generator hands out its CAPTURED state (device pointers
    to seed and offset plus an offset inside the graph), the forward kernel unpacks it on the device
    at replay time (flash_api.cpp:388-394, flash_fwd_kernel.h:45) and writes {seed, offset} to
    rng_state for the backward.  Every replay draws a fresh mask; each replay's results equal an
    eager call with the seed / offset that replay reported, bit for bit."""
    import cabi
    import vllm_flash_attn_2_cuda as ext
    torch.manual_seed(5)
    b, s, h, hk, d, pd = 2, 300, 4, 2, 128, 0.1
    scale = d ** -0.5
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)

    def step():
        res = ext.fwd(q, k, v, None, None, pd, scale, True, -1, -1, False, None)
        out, lse, rng = res[0], res[5], res[7]
        dq, dk, dv, _ = ext.bwd(do, q, k, v, out, lse, None, None, None, None, pd, scale, True, -1, -1, False, None, rng)
        return out, lse, rng, dq, dk, dv

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = step()
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        out, lse, rng, dq, dk, dv = (t.clone() for t in cap)
        seed, off = int(rng[0].item()), int(rng[1].item())
        seen.append((seed, off))
        eo, el = cabi.fwd(q, k, v, causal=True, p_dropout=pd, philox_seed=seed, philox_offset=off)
        assert torch.equal(eo, out) and torch.equal(el, lse)
        edq, edk, edv, _ = cabi.bwd(do, q, k, v, out, lse, causal=True, p_dropout=pd, philox_seed=seed, philox_offset=off)
        assert torch.equal(edq, dq) and torch.equal(edk, dk) and torch.equal(edv, dv)
    assert len(set(seen)) == 3, seen            # every replay advanced the generator
    assert all(b_[1] - a_[1] >= b * h * 32 for a_, b_ in zip(seen, seen[1:]))
