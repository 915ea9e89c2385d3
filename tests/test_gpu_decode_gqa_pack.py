"""Few query tokens per sequence with grouped query heads (speculative decoding, small chunks, ALiBi
decode): the decode kernel packs the query heads of a KV group into the rows of one workgroup.
Checked against the unpacked kernel (same inputs, packing switched off) and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cabi():
    import cabi
    return cabi


def _both(fn):
    with _cabi().dev(2, 2):   # development library, packing off
        plain = fn()
    return plain, fn()


def _oracle_check(q, kc, vc, lens, o, lse, dtype, **kw):
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    f = lambda t: t.float().cpu().numpy()
    name = "bf16" if dtype == torch.bfloat16 else "fp16"
    ro, rl = fa_oracle.attention_fwd(f(q), f(kc), f(vc), dtype=name, cache_seqlens=lens.cpu().numpy(), **kw)
    assert np.abs(f(o) - ro).max() <= ulp_tol(name, ro)
    got = lse.cpu().numpy()
    inf = np.isposinf(rl)
    assert (np.isposinf(got) == inf).all()
    assert np.abs(got[~inf] - rl[~inf]).max(initial=0.0) <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [128, 64, 256, 80])
@pytest.mark.parametrize("sq,h,hk,splits,causal", [(1, 8, 2, 1, True), (2, 32, 8, 1, True), (4, 32, 8, 3, True),
                                                   (5, 12, 4, 1, True), (8, 6, 1, 2, False), (32, 8, 2, 1, True),
                                                   (3, 14, 7, 1, True), (17, 16, 2, 4, True)])
def test_packed_rows_equal_the_unpacked_kernel_and_the_oracle(dtype, d, sq, h, hk, splits, causal):
    torch.manual_seed(sq * 100 + h)
    b, sk = 4, 600
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dtype)
    kc = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    vc = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    lens = torch.tensor([600, 0, sq, 333], dtype=torch.int32, device="cuda")
    (o1, l1), (o2, l2) = _both(lambda: _cabi().fwd(q, kc, vc, cache_seqlens=lens, kvcache=True,
                                                   num_splits=splits, causal=causal))
    # the same keys go through the same tiles, waves and merge for every row: bitwise equal
    assert torch.equal(o1, o2)
    assert torch.equal(l1, l2)
    _oracle_check(q, kc, vc, lens, o2, l2, dtype, causal=causal)


@pytest.mark.parametrize("mask", ["local", "alibi", "alibi_causal"])
def test_packed_rows_masks_and_paged_cache(mask):
    torch.manual_seed(7)
    b, sq, h, hk, d, page = 3, 6, 16, 4, 128, 16
    dtype = torch.float16
    lens = torch.tensor([500, 6, 131], dtype=torch.int32, device="cuda")
    pages_per_seq = 32
    sk = pages_per_seq * page
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dtype)
    kc = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    vc = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    kw, okw = {}, {}
    if mask == "local":
        kw["window"] = okw["window"] = (40, 1)
    if mask == "alibi_causal":
        kw["causal"] = okw["causal"] = True
    if mask.startswith("alibi"):
        slopes = (torch.rand(b, h, device="cuda") * 0.3).float()
        kw["alibi_slopes"] = slopes
        okw["alibi_slopes"] = slopes.cpu().numpy()
    nb = b * pages_per_seq
    perm = torch.randperm(nb, device="cuda")
    bt = perm.view(b, pages_per_seq).to(torch.int32)
    kp = torch.empty(nb, page, hk, d, device="cuda", dtype=dtype)
    vp = torch.empty_like(kp)
    kp[perm] = kc.view(nb, page, hk, d)
    vp[perm] = vc.view(nb, page, hk, d)
    for splits in (1, 3):
        (o1, l1), (o2, l2) = _both(lambda: _cabi().fwd(q, kp, vp, cache_seqlens=lens, block_table=bt,
                                                       kvcache=True, num_splits=splits, **kw))
        assert torch.equal(o1, o2) and torch.equal(l1, l2)
        _oracle_check(q, kc, vc, lens, o2, l2, dtype, **okw)


def test_packed_rows_varlen_and_python_api():
    """Ragged query lengths (flash_attn_varlen_func) and the kv-cache API with appended tokens and
    rotary, grouped heads, few query tokens: packed against unpacked."""
    import vllm_flash_attn as fa
    torch.manual_seed(9)
    h, hk, d = 16, 4, 128
    dtype = torch.bfloat16
    qlens, klens = [3, 1, 7, 2], [300, 40, 7, 129]
    cu_q = torch.tensor([0] + list(np.cumsum(qlens)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(klens)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(qlens), h, d, device="cuda", dtype=dtype)
    k = torch.randn(sum(klens), hk, d, device="cuda", dtype=dtype)
    v = torch.randn(sum(klens), hk, d, device="cuda", dtype=dtype)
    o1, o2 = _both(lambda: fa.flash_attn_varlen_func(q, k, v, cu_q, cu_k, max(qlens), max(klens), causal=True))
    assert torch.equal(o1, o2)

    b, sk, rd, sn = 3, 256, 64, 4
    qq = torch.randn(b, sn, h, d, device="cuda", dtype=dtype)
    kc0 = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    vc0 = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    kn = torch.randn(b, sn, hk, d, device="cuda", dtype=dtype)
    vn = torch.randn(b, sn, hk, d, device="cuda", dtype=dtype)
    lens = torch.tensor([100, 0, 252], dtype=torch.int32, device="cuda")
    ang = torch.rand(sk, rd // 2, device="cuda") * 6.28
    cos, sin = ang.cos().to(dtype), ang.sin().to(dtype)

    def run():
        kc, vc = kc0.clone(), vc0.clone()
        o = fa.flash_attn_with_kvcache(qq, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin,
                                       cache_seqlens=lens, causal=True)
        return o, kc
    (o1, k1), (o2, k2) = _both(run)
    assert torch.equal(o1, o2) and torch.equal(k1, k2)
