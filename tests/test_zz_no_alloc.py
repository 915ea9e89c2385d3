"""GPU test of vfa_mha_fwd_kvcache_noalloc: the variant of vfa_mha_fwd_kvcache that keeps the
split-KV partial results in a static device array of the library instead of caller-provided
scratch (for callers that cannot allocate device memory).  Through the C ABI, against the existing
entry point with caller scratch: same kernels, same split count, so the results must be equal bit
for bit -- on two calls in a row, the second on a smaller input (the array keeps the first call's
contents), then with append / paged caches / the 128-row split kernels / the heuristic count."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(b, sq, sk, h, hk, d, dt, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dt, generator=g)
    kc = torch.randn(b, sk, hk, d, device="cuda", dtype=dt, generator=g)
    vc = torch.randn(b, sk, hk, d, device="cuda", dtype=dt, generator=g)
    lens = torch.randint(sk // 2, sk + 1, (b,), device="cuda", generator=g).to(torch.int32)
    return q, kc, vc, lens


def _both(q, kc, vc, lens, splits, **kw):
    import cabi
    o0, l0 = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=splits, kvcache=True, **kw)
    o1, l1 = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=splits, kvcache=True, noalloc=True, **kw)
    return (o0, l0), (o1, l1)


def test_noalloc_equals_caller_scratch_twice_in_a_row_second_call_smaller():
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    # first call: decode, 4 splits
    q, kc, vc, lens = _mk(6, 1, 3000, 8, 2, 128, torch.float16, 1)
    (o0, l0), (o1, l1) = _both(q, kc, vc, lens, 4, causal=True)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(kc), f32(vc), dtype="fp16", causal=True,
                                     cache_seqlens=lens.cpu().numpy())
    assert np.abs(f32(o1) - ro).max() <= ulp_tol("fp16", ro)
    assert np.abs(f32(l1) - rl).max() <= 1e-3
    # second call, smaller in every dimension and with fewer splits: stale partials of the first
    # call lie behind and between the regions this one uses
    q, kc, vc, lens = _mk(2, 3, 700, 4, 4, 64, torch.float16, 2)
    (o0, l0), (o1, l1) = _both(q, kc, vc, lens, 3, causal=True)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(kc), f32(vc), dtype="fp16", causal=True,
                                     cache_seqlens=lens.cpu().numpy())
    assert np.abs(f32(o1) - ro).max() <= ulp_tol("fp16", ro)


def test_noalloc_other_paths_and_heuristic_count():
    import cabi
    # more than 64 query rows: the 128-row split kernels + combine
    q, kc, vc, lens = _mk(1, 200, 4096, 8, 2, 128, torch.bfloat16, 3)
    lens[:] = 4096
    (o0, l0), (o1, l1) = _both(q, kc, vc, lens, 2, causal=True)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    # append new keys (in-place cache update) with a split
    q, kc, vc, lens = _mk(3, 2, 512, 4, 2, 128, torch.float16, 4)
    lens = torch.tensor([100, 300, 7], dtype=torch.int32, device="cuda")
    kn = torch.randn(3, 2, 2, 128, device="cuda", dtype=torch.float16)
    vn = torch.randn_like(kn)
    ka, va, kb, vb = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    o0, l0 = cabi.fwd(q, ka, va, cache_seqlens=lens, knew=kn, vnew=vn, num_splits=2, causal=True)
    o1, l1 = cabi.fwd(q, kb, vb, cache_seqlens=lens, knew=kn, vnew=vn, num_splits=2, causal=True, noalloc=True)
    assert torch.equal(o0, o1) and torch.equal(l0, l1) and torch.equal(ka, kb) and torch.equal(va, vb)
    # num_splits = 0: the library's own count (1 batch x 8 heads against 8192 keys -> several splits)
    q, kc, vc, lens = _mk(1, 1, 8192, 8, 8, 128, torch.float16, 5)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = cabi.lib().vfa_splitkv_num_splits(1, 8, 128, 8192, 1, cus)
    assert n > 1
    o0, l0 = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=n, kvcache=True)
    o1, l1 = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=0, kvcache=True, noalloc=True)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    # a single split needs no scratch at all
    o2, l2 = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=1, kvcache=True, noalloc=True)
    o3, l3 = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=1, kvcache=True)
    assert torch.equal(o2, o3) and torch.equal(l2, l3)


def test_noalloc_rejects_sizes_beyond_the_static_array_and_allocates_nothing():
    import cabi
    b, sq, sk, h, hk, d, splits = 64, 64, 256, 32, 8, 128, 2
    need = splits * b * h * sq * (d + 1) * 4
    assert need > cabi.NOALLOC_SCRATCH_BYTES
    q, kc, vc, lens = _mk(b, sq, sk, h, hk, d, torch.float16, 6)
    with pytest.raises(cabi.VfaError, match="exceed the static scratch"):
        cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=splits, kvcache=True, noalloc=True)
    # the heuristic count is lowered to what fits instead (here: no split)
    cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=0, kvcache=True, noalloc=True)
    # device memory in use does not move across calls (torch's allocator is not involved either:
    # the outputs are allocated before the measurement)
    q, kc, vc, lens = _mk(4, 1, 2048, 8, 2, 128, torch.float16, 7)
    out = torch.empty_like(q)
    cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=4, kvcache=True, noalloc=True, out=out)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    reserved0 = torch.cuda.memory_reserved()
    for _ in range(3):
        P_out, _ = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=4, kvcache=True, noalloc=True, out=out)
    torch.cuda.synchronize()
    # (cabi.fwd allocates the small LSE tensor from torch's cache; the driver-level figure is the test)
    assert torch.cuda.memory_reserved() == reserved0
    assert torch.cuda.mem_get_info()[0] == free0


def test_noalloc_call_under_graph_capture():
    """The call this entry point exists for: captured into a HIP graph (after one eager call on the
    device, which resolves the array's address) and replayed on new inputs."""
    import cabi
    q, kc, vc, lens = _mk(4, 1, 2048, 8, 2, 128, torch.float16, 8)
    out = torch.empty_like(q)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=4, kvcache=True, noalloc=True, out=out)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=4, kvcache=True, noalloc=True, out=out)
    q.copy_(torch.randn_like(q))
    kc.copy_(torch.randn_like(kc))
    g.replay()
    torch.cuda.synchronize()
    want, _ = cabi.fwd(q, kc, vc, cache_seqlens=lens, num_splits=4, kvcache=True)
    assert torch.equal(out, want)
