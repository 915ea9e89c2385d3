"""A few query positions per sequence with grouped query heads: the prefill kernel with the group
packed into the rows of its KV head (fwd_prefill_pack.hip; what more than 32 packed rows and fewer
than 128 positions take), against the oracle and against what the same call takes with the packing
switched off (development library, switch 2 = 4: 32-row decode blocks or one block per query head)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _f(t):
    return t.float().cpu().numpy()


def _check(out, lse, ro, rl, dt):
    from gpu_util import ulp_tol
    assert not torch.isnan(out).any()
    assert np.abs(_f(out) - ro).max() <= ulp_tol(dt, ro)
    got = lse.cpu().numpy()
    inf = np.isposinf(rl)
    assert (np.isposinf(got) == inf).all()
    assert np.abs(got[~inf] - rl[~inf]).max(initial=0.0) <= 1e-3


DENSE = [
    # dt, b, sq, sk, h, hk, d, causal, window
    ("bf16", 2, 16, 700, 8, 2, 128, True, (-1, -1)),      # 64 packed rows: half a block
    ("fp16", 3, 33, 300, 8, 1, 128, True, (-1, -1)),      # MQA x 8: 264 rows, three blocks, the last one partial
    ("bf16", 1, 100, 100, 6, 2, 128, True, (-1, -1)),     # group of 3: row -> (position, head) by a true division
    ("bf16", 2, 20, 1000, 4, 2, 64, False, (-1, -1)),     # d = 64
    ("fp16", 1, 127, 500, 4, 2, 96, True, (-1, -1)),      # padded head dim, 254 rows
    ("bf16", 2, 40, 900, 8, 2, 128, False, (200, 50)),    # window on both sides
    ("bf16", 1, 50, 20, 8, 2, 128, True, (-1, -1)),       # more positions than keys: the first ones see nothing
    ("fp16", 2, 9, 129, 16, 2, 128, True, (64, 0)),       # 72 rows, causal window
    ("bf16", 2, 24, 800, 8, 2, 256, True, (-1, -1)),      # d = 256
    ("fp16", 1, 70, 333, 4, 1, 192, False, (-1, -1)),     # d = 192 on the 256-column kernel, three blocks
]


@pytest.mark.parametrize("case", DENSE, ids=[str(c) for c in DENSE])
def test_packed_rows_dense_vs_oracle_and_unpacked(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT, ulp_tol
    dt, b, sq, sk, h, hk, d, causal, window = case
    torch.manual_seed(sq * 31 + sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    kw = dict(causal=causal, window=window)
    assert cabi.lib().vfa_dispatch_geometry is not None
    out, lse = cabi.fwd(q, k, v, out=torch.full_like(q, float("nan")), **kw)[:2]
    ro, rl = fa_oracle.attention_fwd(_f(q), _f(k), _f(v), dtype=dt, **kw)
    _check(out, lse, ro, rl, dt)
    with cabi.dev(2, 4):
        o0, l0 = cabi.fwd(q, k, v, **kw)[:2]
    assert np.abs(_f(out) - _f(o0)).max() <= 2 * ulp_tol(dt, ro)
    # and with the keys split over workgroups (fp32 partials at the unpacked (head, position) places)
    o2, l2 = cabi.fwd(q, k, v, num_splits=3, **kw)[:2]
    _check(o2, l2, ro, rl, dt)


def test_packed_rows_ragged_batch():
    import cabi
    from oracle import fa_oracle
    torch.manual_seed(4)
    ql, kl = [40, 9, 1, 120, 17], [300, 9, 77, 120, 1000]
    h, hk, d = 8, 2, 128
    cu = lambda l: torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    cq, ck = cu(ql), cu(kl)
    out, lse = cabi.fwd(q, k, v, causal=True, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))[:2]
    ro, rl = fa_oracle.attention_fwd(_f(q), _f(k), _f(v), dtype="bf16", causal=True, cu_seqlens_q=cq.cpu().numpy(),
                                     cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    from gpu_util import ulp_tol
    assert np.abs(_f(out) - ro).max() <= ulp_tol("bf16", ro)
    for i, n in enumerate(ql):
        assert np.abs(lse.cpu().numpy()[i, :, :n] - rl[i, :, :n]).max() <= 1e-3


@pytest.mark.parametrize("paged", [False, True])
@pytest.mark.parametrize("append", [False, True])
@pytest.mark.parametrize("sq,h,hk,splits", [(16, 8, 2, 1), (5, 16, 2, 1), (32, 8, 2, 4), (64, 4, 2, 2)])
def test_packed_rows_kvcache_call(paged, append, sq, h, hk, splits):
    """flash_attn_with_kvcache with several new positions per sequence: per-sequence key counts, a
    paged cache or a batch-index indirection, new keys appended first (no rotary: rotary tables keep
    the decode kernel, which rotates the query)."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(sq + h + paged)
    b, cap, d, page = 3, 640, 128, 16
    lens = torch.tensor([600 - sq, 37, 256], dtype=torch.int32, device="cuda")
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, cap, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    kn = torch.randn(b, sq, hk, d, device="cuda", dtype=torch.float16) if append else None
    vn = torch.randn_like(kn) if append else None
    ke, ve = kc.clone(), vc.clone()
    tot = lens.clone()
    if append:
        for i in range(b):
            n = int(lens[i])
            ke[i, n:n + sq] = kn[i]
            ve[i, n:n + sq] = vn[i]
        tot = lens + sq
    ro, _ = fa_oracle.attention_fwd(_f(q), _f(ke), _f(ve), dtype="fp16", causal=True, cache_seqlens=tot.cpu().numpy())
    if paged:
        nblk = b * cap // page
        perm = torch.randperm(nblk, device="cuda")
        bt = perm.view(b, cap // page).to(torch.int32)
        kp = torch.empty(nblk, page, hk, d, device="cuda", dtype=torch.float16)
        vp = torch.empty_like(kp)
        kp[perm] = kc.view(nblk, page, hk, d)
        vp[perm] = vc.view(nblk, page, hk, d)
        out = cabi.fwd(q, kp, vp, causal=True, cache_seqlens=lens, block_table=bt, knew=kn, vnew=vn, kvcache=True,
                       num_splits=splits)[0]
    else:
        idx = torch.tensor([2, 0, 1], dtype=torch.int32, device="cuda")
        kcc, vcc = torch.empty_like(kc), torch.empty_like(vc)
        kcc[idx.long()] = kc
        vcc[idx.long()] = vc
        out = cabi.fwd(q, kcc, vcc, causal=True, cache_seqlens=lens, cache_batch_idx=idx, knew=kn, vnew=vn,
                       kvcache=True, num_splits=splits)[0]
    assert not torch.isnan(out).any()
    assert np.abs(_f(out) - ro).max() <= ulp_tol("fp16", ro)


def test_python_api_takes_the_planner_and_the_packed_kernel():
    import vllm_flash_attn as fa
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(2)
    b, sq, kv, h, hk, d = 2, 24, 3000, 32, 8, 128
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    lens = torch.tensor([kv, 1234], dtype=torch.int32, device="cuda")
    out = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True)
    ro, _ = fa_oracle.attention_fwd(_f(q), _f(kc), _f(vc), dtype="bf16", causal=True, cache_seqlens=lens.cpu().numpy())
    assert np.abs(_f(out) - ro).max() <= ulp_tol("bf16", ro)


@pytest.mark.parametrize("paged", [False, True])
def test_mixed_serving_batch_short_on_average(paged):
    """One ragged call with many one-position sequences and a few long chunks (what a serving engine
    batches): fewer than 128 query rows per sequence on average, so the grouped heads are packed
    (the total_q hint of vfa_params), also for the sequences of several 128-row blocks."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(12 + paged)
    ql = [1] * 9 + [300, 1, 130, 2, 1]
    kl = [500, 33, 1000, 64, 65, 700, 16, 129, 400] + [300, 900, 500, 77, 1]
    h, hk, d, page = 8, 2, 128, 16
    cu = lambda l: torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device="cuda")
    cq, ck = cu(ql), cu(kl)
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    assert sum(ql) < 128 * len(ql) and max(ql) >= 128
    kw = dict(causal=True, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    ro, rl = fa_oracle.attention_fwd(_f(q), _f(k), _f(v), dtype="bf16", causal=True, cu_seqlens_q=cq.cpu().numpy(),
                                     cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    if paged:
        b = len(kl)
        maxblk = (max(kl) + page - 1) // page
        nblk = sum((n + page - 1) // page for n in kl) + 2
        kp = torch.full((nblk, page, hk, d), float("nan"), device="cuda", dtype=torch.bfloat16)
        vp = torch.full_like(kp, float("nan"))
        bt = torch.full((b, maxblk), 2 ** 30, dtype=torch.int32, device="cuda")
        perm = torch.randperm(nblk, device="cuda")
        o = 0
        for i, n in enumerate(kl):
            for j in range((n + page - 1) // page):
                rows = min(page, n - j * page)
                pg = int(perm[o]); o += 1
                kp[pg, :rows] = k[int(ck[i]) + j * page: int(ck[i]) + j * page + rows]
                vp[pg, :rows] = v[int(ck[i]) + j * page: int(ck[i]) + j * page + rows]
                bt[i, j] = pg
        out, lse = cabi.fwd(q, kp, vp, block_table=bt, **kw)[:2]
    else:
        out, lse = cabi.fwd(q, k, v, **kw)[:2]
    assert not torch.isnan(out).any()
    assert np.abs(_f(out) - ro).max() <= ulp_tol("bf16", ro)
    for i, n in enumerate(ql):
        assert np.abs(lse.cpu().numpy()[i, :, :n] - rl[i, :, :n]).max() <= 1e-3
    with cabi.dev(2, 4):   # one block per query head
        o0 = (cabi.fwd(q, kp, vp, block_table=bt, **kw) if paged else cabi.fwd(q, k, v, **kw))[0]
    assert np.abs(_f(out) - _f(o0)).max() <= 2 * ulp_tol("bf16", ro)


@pytest.mark.parametrize("causal,window", [(True, (-1, -1)), (False, (-1, -1)), (False, (100, 20)), (True, (64, 0))])
def test_packed_rows_with_alibi(causal, window):
    """ALiBi under the packing: the slope is the one of the row's own head (a per-lane value), in the
    linear form under a causal mask or a window that ends at the diagonal (the row constant goes back
    into the LSE) and in the |i - j| form otherwise."""
    import cabi
    from oracle import fa_oracle
    torch.manual_seed(17)
    b, sq, sk, h, hk, d = 2, 40, 600, 8, 2, 128
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    slopes = torch.rand(b, h, device="cuda") * 0.3
    out, lse = cabi.fwd(q, k, v, causal=causal, window=window, alibi_slopes=slopes)[:2]
    ro, rl = fa_oracle.attention_fwd(_f(q), _f(k), _f(v), dtype="bf16", causal=causal, window=window, alibi_slopes=_f(slopes))
    _check(out, lse, ro, rl, "bf16")
    with cabi.dev(2, 4):
        o0, l0 = cabi.fwd(q, k, v, causal=causal, window=window, alibi_slopes=slopes)[:2]
    from gpu_util import ulp_tol
    assert np.abs(_f(out) - _f(o0)).max() <= 2 * ulp_tol("bf16", ro)
    assert np.abs(lse.cpu().numpy() - l0.cpu().numpy()).max() <= 2e-3


def test_python_varlen_api_passes_the_row_count_hint():
    """flash_attn_varlen_func on a mixed batch: the extension hands q.size(0) to the library
    (vfa_params.total_q), so the call equals the C-ABI call that sets it (packed rows), bit for bit."""
    import cabi
    import vllm_flash_attn as fa
    torch.manual_seed(3)
    ql, kl = [1] * 6 + [200], [300, 64, 1000, 17, 128, 555, 200]
    h, hk, d = 8, 2, 128
    cu = lambda l: torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device="cuda")
    cq, ck = cu(ql), cu(kl)
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.float16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.float16)
    v = torch.randn_like(k)
    out = fa.flash_attn_varlen_func(q, k, v, cq, ck, max(ql), max(kl), causal=True)
    ref = cabi.fwd(q, k, v, causal=True, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))[0]
    assert torch.equal(out, ref)
    with cabi.dev(2, 4):
        unpacked = cabi.fwd(q, k, v, causal=True, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))[0]
    assert (out.float() - unpacked.float()).abs().max().item() <= 2.0 ** -9 * max(1.0, unpacked.float().abs().max().item())
