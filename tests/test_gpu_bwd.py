"""GPU parity tests of the backward path (C ABI and the Python autograd wrappers).

Tolerances are the reference tests' own: max|dx - dx_ref| <= 2 * max|dx_pt - dx_ref| for dense
batches (tests/test_flash_attn.py:1096-1098), 3x for ragged batches (:1402-1404), where dx_pt is
the same-dtype PyTorch gradient stored with the fixture; plus the C oracle to a few ulp."""
import numpy as np
import pytest
import torch

from util import golden_names, load_golden

pytestmark = pytest.mark.gpu

GRAD_CASES = [n for n in golden_names("dense") if "fp32" not in n]


def _cu(lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")


def _inputs(g):
    from gpu_util import to_dev
    m = g["meta"]
    q, k, v, do = (to_dev(g[x], m["dtype"]) for x in "qkvg")
    slopes = torch.from_numpy(g["alibi_slopes"]).cuda() if "alibi_slopes" in g else None
    return m, q, k, v, do, slopes


@pytest.mark.parametrize("via", ["cabi", "py"])
@pytest.mark.parametrize("name", GRAD_CASES)
def test_bwd_vs_golden(name, via):
    import cabi
    import vllm_flash_attn as fa
    g = load_golden(name)
    if "dq_ref" not in g:
        pytest.skip("fixture has no gradients")
    m, q, k, v, do, slopes = _inputs(g)
    kw = dict(causal=m["causal"], window=tuple(m["window"]), alibi_slopes=slopes)
    if m.get("qlens"):
        ql, kl = m["qlens"], m["klens"]
        cat = lambda x, ls: torch.cat([x[i, :n] for i, n in enumerate(ls)])
        qu, dou, ku, vu = cat(q, ql), cat(do, ql), cat(k, kl), cat(v, kl)
        cq, ck = _cu(ql), _cu(kl)
        if via == "cabi":
            vkw = dict(cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))
            out, lse = cabi.fwd(qu, ku, vu, **kw, **vkw)
            dqu, dku, dvu, _ = cabi.bwd(dou, qu, ku, vu, out, lse, **kw, **vkw)
        else:
            qu, ku, vu = (x.clone().requires_grad_() for x in (qu, ku, vu))
            out = fa.flash_attn_varlen_func(qu, ku, vu, cq, ck, max(ql), max(kl), causal=m["causal"],
                                            window_size=tuple(m["window"]), alibi_slopes=slopes)
            dqu, dku, dvu = torch.autograd.grad(out, (qu, ku, vu), dou)
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        for i in range(m["b"]):
            dq[i, :ql[i]] = dqu[cq[i]:cq[i + 1]]
            dk[i, :kl[i]] = dku[ck[i]:ck[i + 1]]
            dv[i, :kl[i]] = dvu[ck[i]:ck[i + 1]]
        mult = 3.0
    else:
        if via == "cabi":
            out, lse = cabi.fwd(q, k, v, **kw)
            dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, **kw)
        else:
            q, k, v = (x.clone().requires_grad_() for x in (q, k, v))
            out = fa.flash_attn_func(q, k, v, causal=m["causal"], window_size=tuple(m["window"]),
                                     alibi_slopes=slopes)
            dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
        mult = 2.0
    add = 1e-5
    if "_x5_" in name or "_x3_" in name:  # overflow inputs: 5x + 1e-3 (tests/test_flash_attn.py:2220-2228)
        mult, add = 5.0, 1e-3
    for nm, a in (("dq", dq), ("dk", dk), ("dv", dv)):
        ref = g[nm + "_ref"].copy()
        if m.get("qlens"):
            lens = m["qlens"] if nm == "dq" else m["klens"]
            for i, n in enumerate(lens):
                ref[i, n:] = 0
        err = np.abs(a.float().cpu().numpy() - ref).max()
        tol = mult * float(g["err_" + nm + "_pt"]) + add
        assert err <= tol, (name, nm, err, tol)


BWD_CASES = [
    ("bf16", 2, 384, 384, 4, 4, 128, True, (-1, -1), False),
    ("fp16", 1, 300, 555, 4, 2, 64, True, (-1, -1), False),
    ("bf16", 1, 512, 512, 4, 1, 128, False, (-1, -1), False),
    ("bf16", 2, 256, 256, 2, 2, 128, True, (64, 0), True),
    ("fp16", 1, 129, 257, 2, 2, 128, False, (30, 40), False),
    ("bf16", 1, 100, 100, 2, 2, 96, True, (-1, -1), False),
    ("bf16", 1, 70, 130, 1, 1, 256, True, (-1, -1), False),
]


@pytest.mark.parametrize("case", BWD_CASES, ids=[str(c) for c in BWD_CASES])
def test_bwd_vs_oracle_random(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    torch.manual_seed(0)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window)
    out, lse = cabi.fwd(q, k, v, alibi_slopes=slopes, **kw)
    dq, dk, dv, sd = cabi.bwd(do, q, k, v, out, lse, alibi_slopes=slopes, **kw)
    f = lambda t: t.float().cpu().numpy()
    rdq, rdk, rdv, rsd = fa_oracle.attention_bwd(f(do), f(q), f(k), f(v), f(out), f(lse), dtype=dt,
                                                 alibi_slopes=None if slopes is None else f(slopes), **kw)
    eps = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    # P and dS are rounded to 16 bit before the products in both implementations; a different fp32
    # summation order can flip such a rounding, so the bound is a few output ulps.
    for nm, a, r in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        err = np.abs(f(a) - r).max()
        assert err <= 4 * eps * max(1.0, np.abs(r).max()), (nm, err, np.abs(r).max())
    assert np.abs(f(sd)[:, :, :sq] - rsd).max() <= 1e-3 * max(1.0, np.abs(rsd).max())


def test_bwd_noncontiguous_dout_and_packed():
    """Non-contiguous dout honoured through its strides (tests/test_flash_attn.py:2239-2283) and
    the packed variants' backward (which cannot run in the reference: wrong gradient arity)."""
    import vllm_flash_attn as fa
    torch.manual_seed(1)
    b, s, h, d = 2, 200, 4, 64
    qkv = torch.randn(b, s, 3, h, d, device="cuda", dtype=torch.float16, requires_grad=True)
    out = fa.flash_attn_qkvpacked_func(qkv, causal=True)
    g_big = torch.randn(b, s, h, 2 * d, device="cuda", dtype=torch.float16)
    g = g_big[..., :d]  # non-contiguous rows, unit last stride
    (dqkv,) = torch.autograd.grad(out, qkv, g)
    q, k, v = (qkv[:, :, i].detach().clone().requires_grad_() for i in range(3))
    out2 = fa.flash_attn_func(q, k, v, causal=True)
    dq, dk, dv = torch.autograd.grad(out2, (q, k, v), g.contiguous())
    assert torch.equal(dqkv[:, :, 1], dk) and torch.equal(dqkv[:, :, 2], dv)
    assert torch.equal(dqkv[:, :, 0], dq)  # no atomics anywhere: dq is bitwise reproducible too
    kv = torch.stack([k.detach(), v.detach()], dim=2).requires_grad_()
    out3 = fa.flash_attn_kvpacked_func(q.detach().requires_grad_(), kv, causal=True)
    (dkv,) = torch.autograd.grad(out3, kv, g)
    assert torch.equal(dkv[:, :, 0], dk) and torch.equal(dkv[:, :, 1], dv)


def test_bwd_dk_dv_repeatable_and_no_nan_on_literal_varlen():
    """dq / dk / dv are all summed in registers in a fixed order (no atomics): bitwise repeatable.
    Literal cu_seqlens of tests/test_flash_attn.py:2291-2316: no NaN."""
    import cabi
    torch.manual_seed(2)
    for cu in ([0, 76, 110, 256], [0, 1, 2, 3]):
        cq = torch.tensor(cu, dtype=torch.int32, device="cuda")
        n = cu[-1]
        q = torch.randn(n, 4, 64, device="cuda", dtype=torch.float16)
        k = torch.randn(n, 2, 64, device="cuda", dtype=torch.float16)
        v = torch.randn(n, 2, 64, device="cuda", dtype=torch.float16)
        do = torch.randn_like(q)
        mx = max(cu[i + 1] - cu[i] for i in range(len(cu) - 1))
        kw = dict(causal=True, cu_seqlens_q=cq, cu_seqlens_k=cq, max_seqlen_q=mx, max_seqlen_k=mx)
        out, lse = cabi.fwd(q, k, v, **kw)
        r0 = cabi.bwd(do, q, k, v, out, lse, **kw)
        for t in r0[:3]:
            assert not torch.isnan(t).any()
        for _ in range(5):
            r1 = cabi.bwd(do, q, k, v, out, lse, **kw)
            assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1]) and torch.equal(r0[2], r1[2])


def test_bwd_overflow_inputs():
    """Inputs scaled x5 / x3 at tiny sequence lengths (tests/test_flash_attn.py:2183-2228): gradients
    stay finite and agree with the oracle.  The reference's own 5x-PyTorch-error rule for such
    inputs is applied to the *_x5_* / *_x3_* golden fixtures in test_bwd_vs_golden."""
    import cabi
    from oracle import fa_oracle
    f = lambda t: t.float().cpu().numpy()
    for s in (1, 2, 5, 17, 128):
        torch.manual_seed(0)
        q = torch.randn(2, s, 4, 64, device="cuda", dtype=torch.float16) * 5
        k, v = [torch.randn(2, s, 4, 64, device="cuda", dtype=torch.float16) * 3 for _ in range(2)]
        g = torch.randn_like(q)
        out, lse = cabi.fwd(q, k, v, causal=True)
        dq, dk, dv, _ = cabi.bwd(g, q, k, v, out, lse, causal=True)
        rdq, rdk, rdv, _ = fa_oracle.attention_bwd(f(g), f(q), f(k), f(v), f(out), f(lse), dtype="fp16", causal=True)
        for a, r in ((dq, rdq), (dk, rdk), (dv, rdv)):
            assert torch.isfinite(a).all()
            assert np.abs(f(a) - r).max() <= 8 * 2.0 ** -11 * max(1.0, np.abs(r).max())


def test_huge_row_stride_takes_the_generic_load_path():
    """K / V whose rows are so far apart that a (batch, head) slice exceeds a 32-bit buffer
    offset: forward and backward switch from buffer loads to plain global loads.  Both paths do the
    same arithmetic, so the results must be bitwise those of the contiguous tensors."""
    import cabi
    torch.manual_seed(5)
    b, sq, sk, h, hk, d = 1, 300, 700, 4, 2, 128
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    wide = 600 * 1024  # elements per row: (sk + 256) rows x 1.2 MB > 2^30 bytes
    kbig = torch.empty(b, sk, wide, device="cuda", dtype=torch.bfloat16)
    vbig = torch.empty(b, sk, wide, device="cuda", dtype=torch.bfloat16)
    ks = kbig[:, :, : hk * d].view(b, sk, hk, d)
    vs = vbig[:, :, : hk * d].view(b, sk, hk, d)
    ks.copy_(k)
    vs.copy_(v)
    assert (sk + 256) * ks.stride(1) * 2 >= 1 << 30
    for causal in (False, True):
        out0, lse0 = cabi.fwd(q, k, v, causal=causal)[:2]
        out1, lse1 = cabi.fwd(q, ks, vs, causal=causal)[:2]
        assert torch.equal(out0, out1) and torch.equal(lse0, lse1)
        g0 = cabi.bwd(do, q, k, v, out0, lse0, causal=causal)
        g1 = cabi.bwd(do, q, ks, vs, out0, lse0, causal=causal)
        for a, c in zip(g0[:3], g1[:3]):
            assert torch.equal(a, c)


@pytest.mark.parametrize("b,h,hk,sq,sk,d", [(8, 4, 1, 300, 300, 128), (9, 2, 1, 520, 520, 64), (3, 12, 3, 257, 400, 128),
                                            (17, 4, 2, 130, 130, 64), (5, 16, 8, 1100, 1100, 128),
                                            (2, 8, 8, 4200, 4200, 128), (40, 1, 1, 64, 700, 128)])
@pytest.mark.parametrize("causal", [True, False])
def test_block_order_covers_every_block(b, h, hk, sq, sk, d, causal):
    """The block -> (batch, head, row / key block) maps take (batch, kv head) pairs in groups per
    XCD with a heaviest-first order across each group: with pair counts that do and do not divide
    by 8, groups that do not divide the pairs of an XCD, and several heads per kv head, every output
    element must still be produced exactly once -- the full batch against one call per batch
    element (bitwise), outputs pre-filled with NaN."""
    import cabi
    torch.manual_seed(b * 100 + h)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    out = torch.full_like(q, float("nan"))
    out, lse = cabi.fwd(q, k, v, causal=causal, out=out)
    dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, causal=causal)
    for t in (out, dq, dk, dv):
        assert not torch.isnan(t).any()
    for i in range(b):
        s_ = slice(i, i + 1)
        o1, l1 = cabi.fwd(q[s_], k[s_], v[s_], causal=causal)
        assert torch.equal(o1, out[s_]) and torch.equal(l1, lse[s_])
        g1 = cabi.bwd(do[s_], q[s_], k[s_], v[s_], o1, l1, causal=causal)
        assert torch.equal(g1[0], dq[s_]) and torch.equal(g1[1], dk[s_]) and torch.equal(g1[2], dv[s_])


@pytest.mark.parametrize("b,h,hk,sq,sk,d,varlen", [(1, 8, 1, 300, 520, 128, False), (2, 6, 2, 257, 257, 64, False),
                                                  (1, 4, 2, 100, 700, 128, True), (1, 16, 4, 1030, 1030, 128, False),
                                                  # d > 128: two workgroups per key block, one per half of the head dim
                                                  (1, 8, 2, 300, 520, 256, False), (1, 4, 2, 100, 391, 192, True)])
@pytest.mark.parametrize("causal", [True, False])
def test_bwd_head_split_matches_unsplit_and_oracle(b, h, hk, sq, sk, d, varlen, causal):
    """Few (batch, kv head) pairs: the heads of a KV group are swept by several workgroups per key
    block and their fp32 partial dK / dV are added by the reduce kernel.  Every split count, also
    the one the library suggests, against the unsplit call (dq bitwise, dk / dv to one rounding of
    the 16-bit result) and the oracle; twice, bitwise equal (slabs are added in index order)."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(h * 10 + hk)
    g = h // hk
    if varlen:
        ql, kl = [sq, sq // 2 + 1], [sk, sk // 3 + 2]
        q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
        kw = dict(causal=causal, cu_seqlens_q=_cu(ql), cu_seqlens_k=_cu(kl), max_seqlen_q=max(ql), max_seqlen_k=max(kl))
        okw = dict(causal=causal, cu_seqlens_q=_cu(ql).cpu().numpy(), cu_seqlens_k=_cu(kl).cpu().numpy(),
                   max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    else:
        q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
        kw, okw = dict(causal=causal), dict(causal=causal)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, **kw)
    dq1, dk1, dv1, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)
    o_dq, o_dk, o_dv, _ = fa_oracle.attention_bwd(f32(do), f32(q), f32(k), f32(v), f32(out), f32(lse), dtype="bf16", **okw)
    for hs in sorted({2, g, None}, key=lambda x: (x is None, x)):
        if hs is not None and hs > g:
            continue
        dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=hs, **kw)
        dq_, dk_, dv_, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=hs, **kw)
        assert torch.equal(dq, dq_) and torch.equal(dk, dk_) and torch.equal(dv, dv_)
        assert torch.equal(dq, dq1)
        for a, a1, o in ((dk, dk1, o_dk), (dv, dv1, o_dv)):
            assert np.abs(f32(a) - f32(a1)).max() <= ulp_tol("bf16", o)
            assert np.abs(f32(a) - o).max() <= 2 * ulp_tol("bf16", o)


def test_headline_shape_bwd_slices_and_linearity():
    """Headline configuration (b8 s4096 h32 d128 bf16 causal, forward + backward).  At b = 1 (batch
    entries are independent, checked bitwise above) two head slices against the oracle; at the
    full size a property that needs no oracle: every gradient is linear in dO and a factor of two
    is exact in both the 16-bit and the fp32 arithmetic, so bwd(2 dO) must equal 2 bwd(dO) bit for
    bit (and softmax_d likewise)."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(3)
    b, s, h, d = 1, 4096, 32, 128
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn_like(q)
    v = torch.randn_like(q)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=True)
    dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, causal=True)
    for hh in (0, 31):
        sl = lambda t: f32(t[:, :, hh:hh + 1])
        o_dq, o_dk, o_dv, _ = fa_oracle.attention_bwd(sl(do), sl(q), sl(k), sl(v), sl(out), f32(lse[:, hh:hh + 1]),
                                                     dtype="bf16", causal=True)
        for got, ref in ((dq, o_dq), (dk, o_dk), (dv, o_dv)):
            assert np.abs(sl(got) - ref).max() <= 2 * ulp_tol("bf16", ref)
    del dq, dk, dv
    b = 8
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn_like(q)
    v = torch.randn_like(q)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=True)
    g1 = cabi.bwd(do, q, k, v, out, lse, causal=True)
    g2 = cabi.bwd(do * 2, q, k, v, out, lse, causal=True)
    for a, c in zip(g1, g2):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a * 2, c)


@pytest.mark.parametrize("causal", [False, True])
def test_every_packed_entry_point_matches_the_unpacked_one(causal):
    """flash_attn_{qkvpacked,kvpacked,varlen_qkvpacked,varlen_kvpacked}_func, forward and autograd
    backward, against flash_attn_func / flash_attn_varlen_func on the same data: the packed
    arguments are strided views of the same elements, so outputs and gradients must be bit-equal
    (the packed gradients come back stacked in the packed layout)."""
    import vllm_flash_attn as fa
    torch.manual_seed(17)
    b, s, h, d = 3, 150, 4, 64
    dt = torch.bfloat16
    qkv = torch.randn(b, s, 3, h, d, device="cuda", dtype=dt)
    g = torch.randn(b, s, h, d, device="cuda", dtype=dt)
    q, k, v = (qkv[:, :, i].detach().clone().requires_grad_() for i in range(3))
    ref = fa.flash_attn_func(q, k, v, causal=causal)
    ref.backward(g)
    p1 = qkv.detach().clone().requires_grad_()
    o1 = fa.flash_attn_qkvpacked_func(p1, causal=causal)
    o1.backward(g)
    assert torch.equal(o1, ref)
    assert torch.equal(p1.grad, torch.stack([q.grad, k.grad, v.grad], dim=2))
    q2 = qkv[:, :, 0].detach().clone().requires_grad_()
    kv2 = qkv[:, :, 1:].detach().clone().requires_grad_()
    o2 = fa.flash_attn_kvpacked_func(q2, kv2, causal=causal)
    o2.backward(g)
    assert torch.equal(o2, ref) and torch.equal(q2.grad, q.grad)
    assert torch.equal(kv2.grad, torch.stack([k.grad, v.grad], dim=2))
    # ragged: three sequences packed back to back
    lens = [150, 77, 1]
    cu = torch.tensor([0, 150, 227, 228], dtype=torch.int32, device="cuda")
    tot = sum(lens)
    pk = torch.randn(tot, 3, h, d, device="cuda", dtype=dt)
    gv = torch.randn(tot, h, d, device="cuda", dtype=dt)
    qv, kv_, vv = (pk[:, i].detach().clone().requires_grad_() for i in range(3))
    rv = fa.flash_attn_varlen_func(qv, kv_, vv, cu, cu, max(lens), max(lens), causal=causal)
    rv.backward(gv)
    p3 = pk.detach().clone().requires_grad_()
    o3 = fa.flash_attn_varlen_qkvpacked_func(p3, cu, max(lens), causal=causal)
    o3.backward(gv)
    assert torch.equal(o3, rv)
    assert torch.equal(p3.grad, torch.stack([qv.grad, kv_.grad, vv.grad], dim=1))
    q4 = pk[:, 0].detach().clone().requires_grad_()
    kv4 = pk[:, 1:].detach().clone().requires_grad_()
    o4 = fa.flash_attn_varlen_kvpacked_func(q4, kv4, cu, cu, max(lens), max(lens), causal=causal)
    o4.backward(gv)
    assert torch.equal(o4, rv) and torch.equal(q4.grad, qv.grad)
    assert torch.equal(kv4.grad, torch.stack([kv_.grad, vv.grad], dim=1))


@pytest.mark.parametrize("window,causal,alibi", [((3100, 0), True, False), ((3500, 40), False, True),
                                                 ((1000, 0), True, True)])
def test_long_sequence_window_paths_vs_oracle(window, causal, alibi):
    """4200 rows under a sliding window, forward and backward against the oracle: windows that span
    3072 keys or more keep the 8-wave forward / dQ workgroups (their window-edge masks and tile
    skipping), shorter ones take the 4-wave ones on the same long sequence."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(5)
    b, s, h, hk, d = 1, 4200, 2, 1, 128
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window, alibi_slopes=slopes)
    okw = dict(dtype="bf16", causal=causal, window=window, alibi_slopes=None if slopes is None else slopes.cpu().numpy())
    out, lse = cabi.fwd(q, k, v, **kw)[:2]
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), **okw)
    assert np.abs(f32(out) - ro).max() <= ulp_tol("bf16", ro)
    assert np.abs(lse.cpu().numpy() - rl).max() <= 1e-3
    dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, **kw)
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(f32(do), f32(q), f32(k), f32(v), ro, rl, **okw)
    for got, ref in ((dq, rdq), (dk, rdk), (dv, rdv)):
        assert np.abs(f32(got) - ref).max() <= 2 * ulp_tol("bf16", ref)


def test_forward_backward_is_graph_capturable():
    """A training step under a HIP graph: forward + autograd backward (small GQA batch, so the
    head-split backward with its fp32 slabs and reduce kernel is part of it) captured once and
    replayed on new inputs written in place -- every call only enqueues work on the capturing
    stream -- against the eager result, bit for bit."""
    import vllm_flash_attn as fa
    torch.manual_seed(23)
    b, s, h, hk, d = 1, 700, 8, 2, 128
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)

    def step():
        out = fa.flash_attn_func(q, k, v, causal=True)
        return (out,) + torch.autograd.grad(out, (q, k, v), do)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        captured = step()
    with torch.no_grad():
        for t in (q, k, v, do):
            t.copy_(torch.randn_like(t))
    g.replay()
    torch.cuda.synchronize()
    eager = step()
    for a, c in zip(captured, eager):
        assert torch.equal(a, c)


RACE_CASES = [(128, 1, 239, False, 0.0), (59, 799, 3, True, 0.0), (64, 257, 257, True, 0.17),
              (128, 1024, 1024, True, 0.0), (96, 1024, 128, False, 0.17), (256, 384, 384, True, 0.0)]


@pytest.mark.parametrize("d,sq,sk,causal,dropout_p", RACE_CASES, ids=[str(c) for c in RACE_CASES])
def test_race_condition_250_runs(d, sq, sk, causal, dropout_p):
    """The reference's race test at its own counts (tests/test_flash_attn.py:2135-2173): batch 60,
    4 heads, fp16, forward + autograd backward 250 times with the same generator seed; out, lse,
    dk, dv must be bitwise equal every time -- and so must dq here (no atomics), where the
    reference only asks for allclose."""
    import vllm_flash_attn as fa
    torch.random.manual_seed(0)
    q = torch.randn(60, sq, 4, d, device="cuda", dtype=torch.float16, requires_grad=True)
    k = torch.randn(60, sk, 4, d, device="cuda", dtype=torch.float16, requires_grad=True)
    v = torch.randn(60, sk, 4, d, device="cuda", dtype=torch.float16, requires_grad=True)
    torch.random.manual_seed(42)
    out0, lse0, _ = fa.flash_attn_func(q, k, v, dropout_p, causal=causal, return_attn_probs=True)
    g = torch.randn_like(out0)
    dq0, dk0, dv0 = torch.autograd.grad(out0, (q, k, v), g)
    for _ in range(250):
        torch.random.manual_seed(42)
        out, lse, _ = fa.flash_attn_func(q, k, v, dropout_p, causal=causal, return_attn_probs=True)
        assert torch.equal(out, out0) and torch.equal(lse, lse0)
        dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
        assert torch.equal(dv, dv0) and torch.equal(dk, dk0) and torch.equal(dq, dq0)


@pytest.mark.parametrize("causal,local", [(False, False), (True, False), (False, True)])
def test_deterministic_backward_50_runs(causal, local):
    """deterministic=True, 50 backward passes bitwise equal (tests/test_flash_attn.py:2349-2375),
    GQA so that dk / dv are sums over the group's heads."""
    import vllm_flash_attn as fa
    torch.random.manual_seed(0)
    b, sq, sk, h, hk, d = 4, 239, 1023, 9, 3, 128
    window = (57, 33) if local else (-1, -1)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    out = fa.flash_attn_func(q, k, v, 0.0, causal=causal, window_size=window, deterministic=True)
    g = torch.randn_like(out)
    dq0, dk0, dv0 = torch.autograd.grad(out, (q, k, v), g, retain_graph=True)
    for _ in range(50):
        dq, dk, dv = torch.autograd.grad(out, (q, k, v), g, retain_graph=True)
        assert torch.equal(dv, dv0) and torch.equal(dk, dk0) and torch.equal(dq, dq0)
