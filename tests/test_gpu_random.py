"""Seeded random sweep of small shapes, forward and backward, HIP kernels (through the C ABI)
against the C oracle: ragged lengths around the tile edges (32 / 64 / 128 / 256), every head dim
class, GQA ratios, causal / window / ALiBi combinations.  Same tolerances as the fixed-case oracle
tests: 2 ulp of the 16-bit output (forward), 4 ulp (gradients)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EDGE = [1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 191, 255, 256, 257, 300, 383, 513]


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        d = int(rng.choice([32, 64, 64, 80, 96, 128, 128, 128, 160, 192, 256]))
        hk = int(rng.choice([1, 2, 3]))
        h = hk * int(rng.choice([1, 1, 2, 4]))
        sq = int(rng.choice(EDGE))
        sk = int(rng.choice(EDGE))
        mode = int(rng.integers(0, 4))  # 0 none, 1 causal, 2 window, 3 causal + left window
        causal = mode in (1, 3)
        window = (-1, -1)
        if mode == 2:
            window = (int(rng.integers(0, 130)), int(rng.integers(0, 70)))
        if mode == 3:
            window = (int(rng.integers(0, 200)), 0)
        out.append((str(rng.choice(["bf16", "fp16"])), int(rng.integers(1, 3)), sq, sk, h, hk, d, causal,
                    window, bool(rng.integers(0, 4) == 0)))
    return out


FWD_CASES = _cases(48, 2024)
BWD_CASES = [c for c in _cases(40, 7) if c[6] <= 128][:28]


def _run(case, with_bwd):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT, ulp_tol
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    torch.manual_seed(sq * 1000 + sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window, alibi_slopes=slopes)
    okw = dict(dtype=dt, causal=causal, window=window,
               alibi_slopes=None if slopes is None else slopes.cpu().numpy())
    qn, kn, vn = (t.float().cpu().numpy() for t in (q, k, v))
    out, lse = cabi.fwd(q, k, v, **kw)[:2]
    ro, rl = fa_oracle.attention_fwd(qn, kn, vn, **okw)
    o = out.float().cpu().numpy()
    assert np.isfinite(o).all()
    assert np.abs(o - ro).max() <= ulp_tol(dt, ro)
    l = lse.cpu().numpy()
    assert np.array_equal(np.isposinf(l), np.isposinf(rl))
    fin = np.isfinite(rl)
    if fin.any():
        assert np.abs(l[fin] - rl[fin]).max() <= 1e-3
    if not with_bwd:
        return
    do = torch.randn_like(q)
    dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, **kw)
    # the oracle differentiates at ITS forward result, like the kernel does at its own
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(do.float().cpu().numpy(), qn, kn, vn, ro, rl, **okw)
    for got, ref in ((dq, rdq), (dk, rdk), (dv, rdv)):
        g = got.float().cpu().numpy()
        assert np.isfinite(g).all()
        assert np.abs(g - ref).max() <= 2 * ulp_tol(dt, ref)


@pytest.mark.parametrize("case", FWD_CASES, ids=[str(c) for c in FWD_CASES])
def test_random_fwd(case):
    _run(case, False)


@pytest.mark.parametrize("case", BWD_CASES, ids=[str(c) for c in BWD_CASES])
def test_random_fwd_bwd(case):
    _run(case, True)


def _varlen_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        d = int(rng.choice([64, 128, 128, 96, 176]))
        hk = int(rng.choice([1, 2]))
        h = hk * int(rng.choice([1, 2, 4]))
        nb = int(rng.integers(2, 5))
        ql = [int(rng.choice(EDGE[:16])) for _ in range(nb)]
        same = bool(rng.integers(0, 2))
        kl = ql if same else [int(rng.choice(EDGE[:16])) for _ in range(nb)]
        causal = bool(rng.integers(0, 2))
        window = (int(rng.integers(0, 100)), 0) if (causal and rng.integers(0, 3) == 0) else (-1, -1)
        out.append((str(rng.choice(["bf16", "fp16"])), tuple(ql), tuple(kl), h, hk, d, causal, window))
    return out


VARLEN_CASES = _varlen_cases(16, 99)


@pytest.mark.parametrize("case", VARLEN_CASES, ids=[str(c) for c in VARLEN_CASES])
def test_random_varlen_fwd_bwd(case):
    """Ragged batches (packed tokens + cu_seqlens), forward and backward, against the oracle."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT, ulp_tol
    dt, ql, kl, h, hk, d, causal, window = case
    torch.manual_seed(sum(ql) * 7 + sum(kl))
    cq = torch.tensor(np.concatenate([[0], np.cumsum(ql)]), dtype=torch.int32, device="cuda")
    ck = torch.tensor(np.concatenate([[0], np.cumsum(kl)]), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(sum(kl), hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn_like(q)
    kw = dict(causal=causal, window=window, cu_seqlens_q=cq, cu_seqlens_k=ck,
              max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    okw = dict(dtype=dt, causal=causal, window=window, cu_seqlens_q=cq.cpu().numpy(),
               cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    qn, kn, vn = (t.float().cpu().numpy() for t in (q, k, v))
    out, lse = cabi.fwd(q, k, v, **kw)[:2]
    ro, rl = fa_oracle.attention_fwd(qn, kn, vn, **okw)
    o = out.float().cpu().numpy()
    assert np.isfinite(o).all() and np.abs(o - ro).max() <= ulp_tol(dt, ro)
    dq, dk, dv, _ = cabi.bwd(do, q, k, v, out, lse, **kw)
    rdq, rdk, rdv, _ = fa_oracle.attention_bwd(do.float().cpu().numpy(), qn, kn, vn, ro, rl, **okw)
    for got, ref in ((dq, rdq), (dk, rdk), (dv, rdv)):
        g = got.float().cpu().numpy()
        assert np.isfinite(g).all()
        assert np.abs(g - ref).max() <= 2 * ulp_tol(dt, ref)
