"""The decode kernel with several heads per workgroup (2 at d = 128, 4 at d = 64: 512 contiguous bytes per key) against the
one-head-per-workgroup geometry: each head goes through the same tiles, waves and merge in both, so
the results must agree bit for bit; one case is also checked against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cabi():
    import cabi
    return cabi


def _both(fn):
    with _cabi().dev(1, 1):   # development library, one head per workgroup
        single = fn()
    return single, fn()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("sq,h,splits,causal", [(1, 8, 1, False), (4, 8, 3, False), (33, 2, 1, True),
                                                (7, 6, 5, True), (1, 32, 2, False)])
@pytest.mark.parametrize("d", [128, 64, 96, 48])
def test_pair_geometry_is_bitwise_equal_to_single_head_geometry(dtype, sq, h, splits, causal, d):
    torch.manual_seed(sq * 100 + h)
    b, sk = 5, 700
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dtype)
    kc = torch.randn(b, sk, h, d, device="cuda", dtype=dtype)
    vc = torch.randn(b, sk, h, d, device="cuda", dtype=dtype)
    lens = torch.tensor([700, 0, 16, 333, 129], dtype=torch.int32, device="cuda")
    (o1, l1), (o2, l2) = _both(lambda: _cabi().fwd(q, kc, vc, cache_seqlens=lens, kvcache=True,
                                                   num_splits=splits, causal=causal))
    assert torch.equal(o1, o2)
    assert torch.equal(l1, l2)
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    f = lambda t: t.float().cpu().numpy()
    name = "bf16" if dtype == torch.bfloat16 else "fp16"
    ro, _ = fa_oracle.attention_fwd(f(q), f(kc), f(vc), dtype=name, cache_seqlens=lens.cpu().numpy(), causal=causal)
    assert np.abs(f(o2) - ro).max() <= ulp_tol(name, ro)


@pytest.mark.parametrize("page", [16, 48])
def test_pair_geometry_paged_alibi_window_batch_index(page):
    torch.manual_seed(page)
    b, sq, h, d = 4, 3, 8, 128
    dtype = torch.float16
    lens = torch.tensor([1000, 17, 512, 255], dtype=torch.int32, device="cuda")
    pages_per_seq = (1000 + page - 1) // page
    sk = pages_per_seq * page
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dtype)
    kc = torch.randn(b, sk, h, d, device="cuda", dtype=dtype)
    vc = torch.randn(b, sk, h, d, device="cuda", dtype=dtype)
    nb = b * pages_per_seq
    perm = torch.randperm(nb, device="cuda")
    bt = perm.view(b, pages_per_seq).to(torch.int32)
    kp = torch.empty(nb, page, h, d, device="cuda", dtype=dtype)
    vp = torch.empty_like(kp)
    kp[perm] = kc.view(nb, page, h, d)
    vp[perm] = vc.view(nb, page, h, d)
    slopes = (torch.rand(b, h, device="cuda") * 0.3).float()
    for kw in (dict(), dict(causal=True, alibi_slopes=slopes), dict(window=(100, 0)), dict(alibi_slopes=slopes)):
        (o1, l1), (o2, l2) = _both(lambda: _cabi().fwd(q, kp, vp, cache_seqlens=lens, block_table=bt,
                                                       kvcache=True, num_splits=2, **kw))
        assert torch.equal(o1, o2) and torch.equal(l1, l2), kw
        (o3, _), (o4, _) = _both(lambda: _cabi().fwd(q, kc, vc, cache_seqlens=lens, kvcache=True, num_splits=1, **kw))
        assert torch.equal(o3, o4), kw
    idx = torch.tensor([3, 0, 2, 1], dtype=torch.int32, device="cuda")
    (o1, _), (o2, _) = _both(lambda: _cabi().fwd(q, kc, vc, cache_seqlens=lens, cache_batch_idx=idx, kvcache=True))
    assert torch.equal(o1, o2)


def test_pair_geometry_append_with_rotary():
    """knew / vnew appended and the query rotated inside the decode kernel, both geometries."""
    import vllm_flash_attn as fa
    torch.manual_seed(3)
    b, sk, h, d, rd = 3, 256, 4, 128, 64
    dtype = torch.bfloat16
    q = torch.randn(b, 1, h, d, device="cuda", dtype=dtype)
    kc0 = torch.randn(b, sk, h, d, device="cuda", dtype=dtype)
    vc0 = torch.randn(b, sk, h, d, device="cuda", dtype=dtype)
    kn = torch.randn(b, 1, h, d, device="cuda", dtype=dtype)
    vn = torch.randn(b, 1, h, d, device="cuda", dtype=dtype)
    lens = torch.tensor([100, 0, 255], dtype=torch.int32, device="cuda")
    ang = torch.rand(sk, rd // 2, device="cuda") * 6.28
    cos, sin = ang.cos().to(dtype), ang.sin().to(dtype)

    def run():
        kc, vc = kc0.clone(), vc0.clone()
        o = fa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin,
                                       cache_seqlens=lens, causal=True)
        return o, kc
    (o1, k1), (o2, k2) = _both(run)
    assert torch.equal(o1, o2) and torch.equal(k1, k2)
