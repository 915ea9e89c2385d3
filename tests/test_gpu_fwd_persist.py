"""The persistent-workgroup forward kernel (fwd_prefill_persist.hip: one workgroup per CU walks a
list of 256-row blocks, the K/V tiles of consecutive blocks form one stream) against the oracle and,
bit for bit, against the one-block-per-workgroup kernel: every wave runs the same instructions on
the same rows in both, so O and LSE must agree exactly.  Development switch 3 = 2 takes the
persistent kernel wherever it can run (the shipped dispatch takes it for large launches only),
3 = 1 never."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _f(t):
    return t.float().cpu().numpy()


def _both(q, k, v, **kw):
    import cabi
    with cabi.dev(3, 2):
        o1, l1 = cabi.fwd(q, k, v, **kw)
    with cabi.dev(3, 1):
        o0, l0 = cabi.fwd(q, k, v, **kw)
    return o1, l1, o0, l0


def _lse_equal(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return np.array_equal(a, b)   # (+inf == +inf)


SMALL = [
    # dt, b, sq, sk, h, hk, d, causal
    ("bf16", 2, 512, 512, 4, 4, 128, True),
    ("bf16", 1, 300, 777, 2, 2, 128, False),     # row and key tails
    ("fp16", 1, 1000, 300, 4, 2, 128, True),     # sq > sk: two whole blocks and part of a third see no key
    ("bf16", 3, 257, 63, 2, 1, 128, False),      # one row past a block edge, a single partial tile
    ("fp16", 2, 640, 900, 8, 2, 96, True),       # padded head dim, grouped heads, sq < sk
    ("bf16", 1, 64, 1000, 2, 2, 128, True),
    ("bf16", 2, 1, 129, 2, 2, 128, False),
    # d <= 64: 4 waves, 128-row blocks, two workgroups per CU
    ("bf16", 2, 512, 512, 4, 4, 64, True),
    ("fp16", 1, 300, 777, 2, 1, 64, False),
    ("bf16", 2, 640, 900, 8, 2, 48, True),       # padded head dim
    ("fp16", 1, 1000, 300, 4, 2, 64, True),      # blocks that see no key
    ("bf16", 3, 129, 63, 2, 1, 32, False),
]


@pytest.mark.parametrize("case", SMALL, ids=[str(c) for c in SMALL])
def test_persistent_forward_vs_oracle_and_block_kernel(case):
    from oracle import fa_oracle
    from gpu_util import TDT, ulp_tol
    dt, b, sq, sk, h, hk, d, causal = case
    torch.manual_seed(sq + 3 * sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    o1, l1, o0, l0 = _both(q, k, v, causal=causal)
    ro, rl = fa_oracle.attention_fwd(_f(q), _f(k), _f(v), dtype=dt, causal=causal)
    assert np.abs(_f(o1) - ro).max() <= ulp_tol(dt, ro)
    got = l1.cpu().numpy()
    inf = np.isposinf(rl)
    assert (np.isposinf(got) == inf).all()
    assert np.abs(got[~inf] - rl[~inf]).max(initial=0.0) <= 1e-3
    assert torch.equal(o1, o0) and _lse_equal(l1, l0)


LARGE = [
    # b, sq, sk, h, hk, causal        (several rounds per workgroup; compared with the block kernel)
    (4, 1024, 1024, 32, 32, True),     # 512 positions: one forward and one reversed round
    (5, 1280, 1280, 13, 13, True),     # 65 (batch, head) pairs: one pair beyond the per-XCD groups
    (2, 2048, 64, 16, 16, False),      # one tile per block: every tile is a seam
    (2, 2304, 2500, 24, 8, True),      # grouped heads, nine blocks per head, key tail
    (8, 777, 777, 20, 4, False),       # partial last block and tile, full attention
    (1, 4096, 4096, 32, 32, True),
]


@pytest.mark.parametrize("case", LARGE, ids=[str(c) for c in LARGE])
def test_persistent_forward_many_rounds_equals_block_kernel(case):
    b, sq, sk, h, hk, causal = case
    torch.manual_seed(sq + h)
    q = torch.randn(b, sq, h, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, 128, device="cuda", dtype=torch.bfloat16)
    o1, l1, o0, l0 = _both(q, k, v, causal=causal)
    assert not torch.isnan(o1).any()
    assert torch.equal(o1, o0) and _lse_equal(l1, l0)
    # output buffers poisoned first: every row of every block must be written by the walk
    import cabi
    out = torch.full_like(q, float("nan"))
    with cabi.dev(3, 2):
        o2, l2 = cabi.fwd(q, k, v, causal=causal, out=out)
    assert torch.equal(o2, o0) and _lse_equal(l2, l0)


LARGE64 = [
    # b, sq, sk, h, hk, d, causal, waves (0 = the shipped 4-wave geometry, 8 = development switch 6 = 1)
    (4, 2048, 2048, 16, 16, 64, True, 0),      # 1024 positions over 512 workgroups
    (4, 2048, 2048, 16, 16, 64, True, 8),
    (8, 640, 700, 12, 4, 64, False, 0),
    (3, 1408, 1408, 11, 11, 40, True, 0),      # 33 pairs, 11 blocks each, padded head dim
    (2, 2048, 64, 16, 16, 64, False, 0),       # one tile per block
    (2, 1000, 3000, 8, 2, 64, True, 8),
]


@pytest.mark.parametrize("case", LARGE64, ids=[str(c) for c in LARGE64])
def test_persistent_forward_small_head_dims_equal_block_kernel(case):
    import cabi
    b, sq, sk, h, hk, d, causal, waves = case
    torch.manual_seed(sq + h + d)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    with cabi.dev(3, 2) as lib:
        lib.vfa_debug_set_tuning(6, 1 if waves == 8 else 0)
        try:
            o1, l1 = cabi.fwd(q, k, v, causal=causal, out=torch.full_like(q, float("nan")))
        finally:
            lib.vfa_debug_set_tuning(6, 0)
    with cabi.dev(3, 1):
        o0, l0 = cabi.fwd(q, k, v, causal=causal)
    assert not torch.isnan(o1).any()
    assert torch.equal(o1, o0) and _lse_equal(l1, l0)


def test_shipped_dispatch_takes_it_for_the_headline_shape_and_strided_inputs():
    """b8 s4096 h32 d128 causal through the product library (no switch) equals the block kernel; q,
    k, v as views of one packed (b, s, 3, h, d) tensor (row stride 3 h d)."""
    import cabi
    torch.manual_seed(1)
    qkv = torch.randn(8, 4096, 3, 32, 128, device="cuda", dtype=torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o, l = cabi.fwd(q, k, v, causal=True)
    with cabi.dev(3, 1):
        o0, l0 = cabi.fwd(q, k, v, causal=True)
    assert torch.equal(o, o0) and _lse_equal(l, l0)


# ---- the persistent dQ kernel (bwd_dq_persist.hip), same switch; and the persistent dK/dV kernel,
# which exists in the development library only (switch 0 = 6: measured behind the block kernel) ----

def _bwd_both(do, q, k, v, out, lse, **kw):
    import cabi
    with cabi.dev(3, 2) as lib:
        lib.vfa_debug_set_tuning(0, 6)
        try:
            g1 = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)[:3]
        finally:
            lib.vfa_debug_set_tuning(0, 0)
    # reference: one block per workgroup, and the 8-wave dK/dV geometry (switch 0 = 1), whose
    # accumulation order over sub-blocks and grouped heads the persistent kernel shares
    with cabi.dev(3, 1) as lib:
        lib.vfa_debug_set_tuning(0, 1)
        try:
            g0 = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)[:3]
        finally:
            lib.vfa_debug_set_tuning(0, 0)
    return g1, g0


@pytest.mark.parametrize("case", SMALL, ids=[str(c) for c in SMALL])
def test_persistent_dq_vs_oracle_and_block_kernel(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT
    dt, b, sq, sk, h, hk, d, causal = case
    torch.manual_seed(sq + 5 * sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    out, lse = cabi.fwd(q, k, v, causal=causal)
    (dq, dk, dv), (dq0, dk0, dv0) = _bwd_both(do, q, k, v, out, lse, causal=causal)
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(_f(do), _f(q), _f(k), _f(v), _f(out), _f(lse), dtype=dt, causal=causal)
    eps = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    for nm, a, r in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        err = np.abs(_f(a) - r).max()
        assert err <= 4 * eps * max(1.0, np.abs(r).max()), (nm, err)
    assert torch.equal(dq, dq0) and torch.equal(dk, dk0) and torch.equal(dv, dv0)


@pytest.mark.parametrize("case", LARGE, ids=[str(c) for c in LARGE])
def test_persistent_dq_many_rounds_equals_block_kernel(case):
    import cabi
    b, sq, sk, h, hk, causal = case
    torch.manual_seed(sq + 2 * h)
    q = torch.randn(b, sq, h, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, 128, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=causal)
    (dq, dk, dv), (dq0, dk0, dv0) = _bwd_both(do, q, k, v, out, lse, causal=causal)
    assert not torch.isnan(dq).any()
    assert torch.equal(dq, dq0) and torch.equal(dk, dk0) and torch.equal(dv, dv0)
