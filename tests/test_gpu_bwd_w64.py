"""The dK/dV kernel with 64 keys per wave at one wave per SIMD (bwd_dkdv_w64.hip): what the shipped
dispatch takes for 64 < d <= 128 without a mask once both sequence lengths reach 3072 and the
launch has a 256-key block for every CU.  Checked against the oracle and, bit for bit, against the
8-wave kernel (development library, switch 0 = 1): on the shipped dispatch (product library, no
switch) for a launch of that size, and forced (switch 0 = 2) on smaller launches and on its masked
and d <= 64 instantiations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cu(lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")


def _f(t):
    return t.float().cpu().numpy()


def _check_vs_oracle(dt, got, ref):
    eps = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    for nm, a, r in zip(("dq", "dk", "dv"), got, ref):
        err = np.abs(_f(a) - r).max()
        assert err <= 4 * eps * max(1.0, np.abs(r).max()), (nm, err, np.abs(r).max())


DEFAULT_CASES = [
    # dt, b, sq, sk, h, hk, d, alibi      (no mask, both lengths >= 2048)
    ("bf16", 8, 3100, 3300, 4, 4, 128, False),     # 416 key blocks: the SHIPPED dispatch takes the 64-key kernel
    ("bf16", 1, 2100, 2300, 2, 2, 128, False),     # row and key tails
    ("fp16", 1, 2048, 2176, 4, 1, 96, False),      # padded head dim, four query heads per KV head
    ("bf16", 1, 2112, 2048, 2, 1, 128, True),      # ALiBi in its |i - j| form
    ("fp16", 2, 2050, 2049, 1, 1, 128, False),     # one row / one key past a block edge
]


@pytest.mark.parametrize("case", DEFAULT_CASES, ids=[str(c) for c in DEFAULT_CASES])
def test_default_dispatch_long_unmasked_vs_oracle_and_8_wave_kernel(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT
    dt, b, sq, sk, h, hk, d, alibi = case
    torch.manual_seed(sq + sk + d)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    out, lse = cabi.fwd(q, k, v, alibi_slopes=slopes)
    # head_splits = 1: with a split the dispatch keeps the 8-wave kernel
    if b * hk * ((sk + 255) // 256) >= 256 and min(sq, sk) >= 3072:   # (tests/test_dispatch.py pins this shape to the 64-key kernel)
        dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, alibi_slopes=slopes, head_splits=1)
    else:
        with cabi.dev(0, 2):
            dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, alibi_slopes=slopes, head_splits=1)
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(_f(do), _f(q), _f(k), _f(v), _f(out), _f(lse), dtype=dt,
                                               alibi_slopes=None if slopes is None else _f(slopes))
    _check_vs_oracle(dt, (dq, dk, dv), (rdq, rdk, rdv))
    with cabi.dev(0, 1):   # the 8-wave x 32-key kernel on the same inputs
        dq8, dk8, dv8, _ = cabi.bwd(do, q, k, v, out, lse, alibi_slopes=slopes, head_splits=1)
    assert torch.equal(dk, dk8) and torch.equal(dv, dv8) and torch.equal(dq, dq8)


def test_ragged_batch_with_long_and_very_short_sequences():
    import cabi
    from oracle import fa_oracle
    torch.manual_seed(11)
    ql, kl = [2100, 37, 1, 300], [2200, 37, 5, 2060]
    h, hk, d = 2, 1, 128
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    cq, ck = _cu(ql), _cu(kl)
    vk = dict(cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    out, lse = cabi.fwd(q, k, v, **vk)
    with cabi.dev(0, 2):
        dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **vk)
    nk = dict(cu_seqlens_q=cq.cpu().numpy(), cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(_f(do), _f(q), _f(k), _f(v), _f(out), _f(lse), dtype="bf16", **nk)
    _check_vs_oracle("bf16", (dq, dk, dv), (rdq, rdk, rdv))
    with cabi.dev(0, 1):
        dq8, dk8, dv8, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **vk)
    assert torch.equal(dk, dk8) and torch.equal(dv, dv8)


FORCED_CASES = [
    # dt, b, sq, sk, h, hk, d, causal, window, alibi   (development switch 0 = 2: masks and d <= 64)
    ("bf16", 2, 384, 520, 2, 2, 128, True, (-1, -1), False),
    ("fp16", 1, 700, 300, 4, 2, 128, True, (-1, -1), True),     # sq > sk: leading rows see nothing
    ("bf16", 1, 513, 777, 2, 2, 128, False, (100, 30), False),
    ("bf16", 2, 300, 300, 4, 1, 64, False, (-1, -1), False),
    ("fp16", 1, 257, 600, 2, 2, 48, True, (-1, -1), False),
]


@pytest.mark.parametrize("case", FORCED_CASES, ids=[str(c) for c in FORCED_CASES])
def test_forced_64_key_kernel_masks_and_small_head_dims(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    torch.manual_seed(sq * 7 + sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window, alibi_slopes=slopes)
    out, lse = cabi.fwd(q, k, v, **kw)
    with cabi.dev(0, 2):
        dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)
    with cabi.dev(0, 1):
        dq8, dk8, dv8, _ = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(_f(do), _f(q), _f(k), _f(v), _f(out), _f(lse), dtype=dt,
                                               causal=causal, window=window,
                                               alibi_slopes=None if slopes is None else _f(slopes))
    _check_vs_oracle(dt, (dq, dk, dv), (rdq, rdk, rdv))
    if d > 64:
        assert torch.equal(dk, dk8) and torch.equal(dv, dv8)
    else:
        # below d = 128 the 8-wave kernel subtracts LSE * log2(e) inside the exponent while this one
        # starts the S accumulator from -LSE / scale: same value, one more rounding, so a P or dS
        # can land on the other side of a 16-bit rounding boundary
        eps = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
        for a, a8 in ((dk, dk8), (dv, dv8)):
            assert np.abs(_f(a) - _f(a8)).max() <= 2 * eps * max(1.0, np.abs(_f(a8)).max())


def test_shipped_dispatch_very_long_causal_sequences_equal_the_8_wave_kernel():
    """From 12288 keys under a causal mask the shipped dispatch takes the 64-key kernel too (its waves
    cannot skip the pieces above the diagonal, which only pays back on long sweeps): same bits as the
    8-wave kernel, whose masks the oracle checks at sizes it can reach."""
    import cabi
    torch.manual_seed(5)
    b, s, h, d = 1, 12400, 11, 128
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn_like(q)
    v = torch.randn_like(q)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=True)
    assert cabi.lib().vfa_dispatch_geometry is not None
    dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, causal=True, head_splits=1)
    with cabi.dev(0, 1):
        dq8, dk8, dv8, _ = cabi.bwd(do, q, k, v, out, lse, causal=True, head_splits=1)
    assert not torch.isnan(dk).any() and not torch.isnan(dv).any()
    assert torch.equal(dk, dk8) and torch.equal(dv, dv8) and torch.equal(dq, dq8)
