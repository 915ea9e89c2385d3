"""The dK/dV kernel with the two-phase schedule (the two waves of a SIMD alternate between a matrix
phase and a vector phase; opt-in through development switch 0 = 3) against the oracle and against
the default 8-wave kernel: shapes around its 256-key blocks and 32-row sub-blocks, every mask it
takes (causal, window, key and row tails), ALiBi, grouped heads, the head split and ragged batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _with_kernel(fn, value=3):
    import cabi
    with cabi.dev(0, value):   # the schedule exists in the development library only
        return fn()


CASES = [
    # dt, b, sq, sk, h, hk, causal, window, alibi
    ("bf16", 2, 384, 384, 4, 4, True, (-1, -1), False),
    ("bf16", 1, 512, 512, 4, 1, False, (-1, -1), False),
    ("bf16", 2, 256, 256, 2, 2, True, (64, 0), True),
    ("fp16", 1, 129, 257, 2, 2, False, (30, 40), False),
    ("bf16", 1, 33, 300, 2, 2, False, (-1, -1), False),      # one full and one ragged sub-block
    ("fp16", 2, 300, 1000, 6, 2, True, (-1, -1), False),     # sq < sk under a causal mask, grouped heads
    ("bf16", 1, 1000, 300, 4, 2, True, (-1, -1), True),      # sq > sk: leading rows see nothing
    ("bf16", 1, 31, 129, 1, 1, False, (-1, -1), False),      # a single ragged sub-block
    ("bf16", 1, 700, 700, 2, 1, False, (100, 30), True),
    ("fp16", 1, 1100, 1100, 2, 2, True, (-1, -1), False),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_two_phase_dkdv_matches_oracle(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT
    dt, b, sq, sk, h, hk, causal, window, alibi = case
    d = 128
    torch.manual_seed(sq * 3 + sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window)
    out, lse = cabi.fwd(q, k, v, alibi_slopes=slopes, **kw)
    dq, dk, dv, sd = _with_kernel(lambda: cabi.bwd(do, q, k, v, out, lse, alibi_slopes=slopes, **kw))
    dq0, dk0, dv0, _ = _with_kernel(lambda: cabi.bwd(do, q, k, v, out, lse, alibi_slopes=slopes, **kw), 4)
    f = lambda t: t.float().cpu().numpy()
    rdq, rdk, rdv, rsd = fa_oracle.attention_bwd(f(do), f(q), f(k), f(v), f(out), f(lse), dtype=dt,
                                                 alibi_slopes=None if slopes is None else f(slopes), **kw)
    eps = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    for nm, a, r in (("dk", dk, rdk), ("dv", dv, rdv), ("dq", dq, rdq)):
        err = np.abs(f(a) - r).max()
        assert err <= 4 * eps * max(1.0, np.abs(r).max()), (nm, err, np.abs(r).max())
    assert torch.equal(dq, dq0)  # the dQ kernel does not depend on the switch
    # against the default kernel: same products, -LSE / scale formed by a multiplication with the
    # reciprocal instead of a division, so a rounding of P or dS to 16 bit can flip
    for nm, a, a0 in (("dk", dk, dk0), ("dv", dv, dv0)):
        assert np.abs(f(a) - f(a0)).max() <= 2 * eps * max(1.0, np.abs(f(a0)).max()), nm


def test_two_phase_dkdv_head_split_and_ragged_batch():
    """Few (batch, KV head, key block) items: the heads of a group are split over workgroups
    (fp32 partial slabs + reduction kernel); and a ragged batch through the varlen entry."""
    import cabi
    from oracle import fa_oracle
    torch.manual_seed(5)
    f = lambda t: t.float().cpu().numpy()
    eps = 2.0 ** -8
    # head split: batch 1, 8 query heads on 1 KV head
    b, sq, sk, h, hk, d = 1, 300, 520, 8, 1, 128
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=True)
    dq, dk, dv, _ = _with_kernel(lambda: cabi.bwd(do, q, k, v, out, lse, causal=True))
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(f(do), f(q), f(k), f(v), f(out), f(lse), dtype="bf16", causal=True)
    for nm, a, r in (("dk", dk, rdk), ("dv", dv, rdv)):
        assert np.abs(f(a) - r).max() <= 4 * eps * max(1.0, np.abs(r).max()), nm
    # ragged batch
    lens = [300, 1, 64, 257, 700]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    h, hk = 4, 2
    q = torch.randn(sum(lens), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(sum(lens), hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    vk = dict(cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=max(lens), max_seqlen_k=max(lens), causal=True)
    out, lse = cabi.fwd(q, k, v, **vk)[:2]
    dq, dk, dv, _ = _with_kernel(lambda: cabi.bwd(do, q, k, v, out, lse, **vk))
    dq0, dk0, dv0, _ = _with_kernel(lambda: cabi.bwd(do, q, k, v, out, lse, **vk), 4)
    for nm, a, a0 in (("dk", dk, dk0), ("dv", dv, dv0)):
        assert np.abs(f(a) - f(a0)).max() <= 2 * eps * max(1.0, np.abs(f(a0)).max()), nm
    assert not torch.isnan(dk).any() and not torch.isnan(dv).any()
