"""GPU parity tests of the head-dim pad path, the wide head dims in the backward, and seqused_k.

Head dims that are not a multiple of 8 (the reference's own parametrisation has 40, 59, 111:
tests/test_flash_attn.py:555) are zero-padded by the extension module to the next multiple, the
kernels run on the padded tensors and the outputs / gradients are sliced back
(csrc/flash_attn/flash_api.cpp:310-319,407-411; upstream's mha_bwd pads dout the same way).  The
oracle takes the unpadded tensors: zero columns change neither the scores nor the other columns of
the outputs, so the comparison is to the usual 2 ulp (forward) / 4 ulp (gradients) of the 16-bit
result, with softmax_scale = d_og ** -0.5 on both sides."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cu(lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")


def _check_grads(got, ref, dt):
    from gpu_util import f32, ulp_tol
    for nm, a, r in zip(("dq", "dk", "dv"), got, ref):
        assert a.shape == r.shape, (nm, a.shape, r.shape)
        assert np.isfinite(f32(a)).all(), nm
        err = np.abs(f32(a) - r).max()
        assert err <= 2 * ulp_tol(dt, r), (nm, err)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [40, 59, 111])
def test_flash_attn_func_odd_head_dim_fwd_bwd(d, causal, dt):
    """flash_attn_func / flash_attn_kvpacked_func at d % 8 != 0: forward, out= copy-back and
    autograd backward against the oracle on the unpadded tensors."""
    import vllm_flash_attn as fa
    from oracle import fa_oracle
    from gpu_util import TDT, f32, ulp_tol
    torch.manual_seed(d)
    b, sq, sk, h, hk = 2, 113, 203, 4, 2
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt], requires_grad=True)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt], requires_grad=True)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt], requires_grad=True)
    g = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    out, lse, _ = fa.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
    assert out.shape == q.shape
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), dtype=dt, causal=causal)
    assert np.abs(f32(out) - ro).max() <= ulp_tol(dt, ro)
    assert np.abs(f32(lse) - rl).max() <= 1e-3
    grads = torch.autograd.grad(out, (q, k, v), g)
    ref = fa_oracle.attention_bwd(f32(g), f32(q), f32(k), f32(v), ro, rl, dtype=dt, causal=causal)[:3]
    _check_grads(grads, ref, dt)
    # caller-provided output buffer with the unpadded head dim: filled by the copy-back
    buf = torch.full_like(q, float("nan")).detach()
    o2 = fa.flash_attn_func(q.detach(), k.detach(), v.detach(), causal=causal, out=buf)
    assert torch.equal(buf, out.detach()) and torch.equal(o2, out.detach())
    # packed kv: gradients come back stacked with the caller's head dim
    kv = torch.stack([k.detach(), v.detach()], dim=2).requires_grad_()
    q2 = q.detach().clone().requires_grad_()
    o3 = fa.flash_attn_kvpacked_func(q2, kv, causal=causal)
    dq3, dkv3 = torch.autograd.grad(o3, (q2, kv), g)
    assert torch.equal(o3, out.detach()) and torch.equal(dq3, grads[0])
    assert torch.equal(dkv3[:, :, 0], grads[1]) and torch.equal(dkv3[:, :, 1], grads[2])


@pytest.mark.parametrize("d", [40, 59, 111])
def test_flash_attn_varlen_func_odd_head_dim_fwd_bwd(d):
    """flash_attn_varlen_func / flash_attn_varlen_qkvpacked_func at d % 8 != 0 (ragged batch)."""
    import vllm_flash_attn as fa
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(100 + d)
    dt, h = "bf16", 3
    lens = [76, 34, 146]
    cu = _cu(lens)
    tot = sum(lens)
    qkv = torch.randn(tot, 3, h, d, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(tot, h, d, device="cuda", dtype=torch.bfloat16)
    q, k, v = (qkv[:, i].detach().clone().requires_grad_() for i in range(3))
    out, lse, _ = fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True,
                                            return_attn_probs=True)
    okw = dict(dtype=dt, causal=True, cu_seqlens_q=cu.cpu().numpy(), cu_seqlens_k=cu.cpu().numpy(),
               max_seqlen_q=max(lens), max_seqlen_k=max(lens))
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), **okw)
    assert out.shape == q.shape
    assert np.abs(f32(out) - ro).max() <= ulp_tol(dt, ro)
    grads = torch.autograd.grad(out, (q, k, v), g)
    ref = fa_oracle.attention_bwd(f32(g), f32(q), f32(k), f32(v), ro, rl, **okw)[:3]
    _check_grads(grads, ref, dt)
    p = qkv.detach().clone().requires_grad_()
    o2 = fa.flash_attn_varlen_qkvpacked_func(p, cu, max(lens), causal=True)
    (dp,) = torch.autograd.grad(o2, p, g)
    assert torch.equal(o2, out.detach())
    assert dp.shape == qkv.shape and torch.equal(dp, torch.stack(grads, dim=1))


def test_extension_bwd_pads_unpadded_inputs_and_fills_given_buffers():
    """`bwd` called the way an external caller might: unpadded q / k / v / out / dout at d = 59 and
    caller-allocated dq / dk / dv of that shape -- equal to the autograd path bit for bit."""
    import vllm_flash_attn as fa
    import vllm_flash_attn_2_cuda as ext
    torch.manual_seed(59)
    b, s, h, d = 1, 90, 2, 59
    q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.float16, requires_grad=True) for _ in range(3))
    g = torch.randn(b, s, h, d, device="cuda", dtype=torch.float16)
    out, lse, _ = fa.flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    want = torch.autograd.grad(out, (q, k, v), g)
    dq, dk, dv = (torch.full_like(q, float("nan")).detach() for _ in range(3))
    res = ext.bwd(g, q.detach(), k.detach(), v.detach(), out.detach(), lse, dq, dk, dv, None, 0.0, d ** -0.5,
                  True, -1, -1, False, None, None)
    for a, r, w in zip((dq, dk, dv), res[:3], want):
        assert a.shape == w.shape and torch.equal(a, w) and torch.equal(r, w)


@pytest.mark.parametrize("d", [40, 111])
def test_fwd_kvcache_odd_head_dim_with_append(d):
    """flash_attn_with_kvcache at d % 8 != 0 with new keys appended: the padded cache copy is
    written back into the caller's cache (flash_api.cpp:924-929), out is sliced."""
    import vllm_flash_attn as fa
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(d)
    b, sq, cap, h, hk, snew = 2, 3, 160, 4, 2, 3
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, cap, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn(b, cap, hk, d, device="cuda", dtype=torch.float16)
    kn = torch.randn(b, snew, hk, d, device="cuda", dtype=torch.float16)
    vn = torch.randn(b, snew, hk, d, device="cuda", dtype=torch.float16)
    lens = torch.tensor([100, 37], dtype=torch.int32, device="cuda")
    k0, v0 = kc.clone(), vc.clone()
    out = fa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, cache_seqlens=lens, causal=True)
    assert out.shape == q.shape
    ke, ve = k0.clone(), v0.clone()
    for i, n in enumerate(lens.tolist()):
        ke[i, n:n + snew] = kn[i]
        ve[i, n:n + snew] = vn[i]
    assert torch.equal(kc, ke) and torch.equal(vc, ve)
    ro, _ = fa_oracle.attention_fwd(f32(q), f32(ke), f32(ve), dtype="fp16", causal=True,
                                    cache_seqlens=(lens + snew).cpu().numpy())
    assert np.abs(f32(out) - ro).max() <= ulp_tol("fp16", ro)


WIDE = [
    ("bf16", 2, 200, 330, 4, 2, 160, True, (-1, -1), False),
    ("fp16", 1, 257, 257, 2, 2, 192, False, (-1, -1), False),
    ("bf16", 1, 129, 300, 3, 1, 224, True, (-1, -1), True),
    ("fp16", 2, 300, 140, 2, 1, 192, False, (40, 25), False),
    ("bf16", 1, 520, 520, 2, 2, 160, True, (100, 0), False),
    ("bf16", 1, 64, 700, 4, 4, 224, False, (-1, -1), False),
    ("fp16", 1, 383, 383, 2, 2, 256, True, (-1, -1), False),
]


@pytest.mark.parametrize("case", WIDE, ids=[str(c) for c in WIDE])
def test_wide_head_dim_fwd_bwd_vs_oracle(case):
    """d in {160, 192, 224, 256}: the DP = 256 instantiations, forward and backward against the
    oracle.  The dK/dV kernel of the shipped dispatch keeps all 256 columns of 32 keys per wave in
    one workgroup (bwd_dkdv_wide.hip); the kernel that cuts the head dim in two column halves
    (what a head split, dropout or 64-bit offsets still take) is checked on the same inputs through
    the development library (switch 0 = 5)."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT, f32, ulp_tol
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    torch.manual_seed(d + sq)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn_like(q)
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window, alibi_slopes=slopes)
    okw = dict(dtype=dt, causal=causal, window=window, alibi_slopes=None if slopes is None else f32(slopes))
    out, lse = cabi.fwd(q, k, v, **kw)[:2]
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), **okw)
    assert np.abs(f32(out) - ro).max() <= ulp_tol(dt, ro)
    grads = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)[:3]
    ref = fa_oracle.attention_bwd(f32(do), f32(q), f32(k), f32(v), ro, rl, **okw)[:3]
    _check_grads(grads, ref, dt)
    with cabi.dev(0, 5):
        halves = cabi.bwd(do, q, k, v, out, lse, head_splits=1, **kw)[:3]
    _check_grads(halves, ref, dt)
    assert torch.equal(grads[0], halves[0])   # (dQ: the same kernel on the same inputs)


def test_wide_head_dim_many_steps_grouped_heads_and_ragged_batch():
    """The one-workgroup dK/dV kernel at d = 256 over several key blocks and many 32-row steps:
    a dense causal batch with grouped heads against the column-halves kernel (the same products
    accumulated in the same order over k-steps, slices and heads: bitwise equal), and a ragged
    batch with very short sequences against the oracle."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import f32
    torch.manual_seed(5)
    b, s, h, hk, d = 2, 1200, 6, 2, 256
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s + 77, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=True)[:2]
    g = cabi.bwd(do, q, k, v, out, lse, causal=True, head_splits=1)[:3]
    with cabi.dev(0, 5):
        g0 = cabi.bwd(do, q, k, v, out, lse, causal=True, head_splits=1)[:3]
    for a, a0 in zip(g, g0):
        assert not torch.isnan(a).any()
        assert torch.equal(a, a0)
    ql, kl = [700, 37, 1, 300], [650, 37, 5, 460]
    h, hk, d = 2, 1, 192
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.float16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.float16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    cq, ck = _cu(ql), _cu(kl)
    vk = dict(cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    out, lse = cabi.fwd(q, k, v, causal=True, **vk)[:2]
    g = cabi.bwd(do, q, k, v, out, lse, causal=True, head_splits=1, **vk)[:3]
    nk = dict(cu_seqlens_q=cq.cpu().numpy(), cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=max(ql), max_seqlen_k=max(kl))
    ref = fa_oracle.attention_bwd(f32(do), f32(q), f32(k), f32(v), f32(out), f32(lse), dtype="fp16", causal=True, **nk)[:3]
    _check_grads(g, ref, "fp16")


WIDE_FWD = [
    # dt, b, sq, sk, h, hk, d, causal, window     (several row blocks and many tiles per block)
    ("bf16", 2, 1200, 1200, 4, 2, 256, True, (-1, -1)),
    ("fp16", 1, 700, 1500, 2, 2, 256, True, (-1, -1)),      # sq < sk: the diagonal starts 800 keys in
    ("bf16", 1, 1500, 600, 2, 1, 192, True, (-1, -1)),      # sq > sk: leading row blocks see no key
    ("bf16", 1, 1000, 1000, 2, 2, 256, False, (300, 100)),  # window: the wave's tile range starts inside the block's
    ("fp16", 3, 129, 67, 2, 2, 256, False, (-1, -1)),       # one row past a block edge, two tiles, key tail
    ("bf16", 1, 1, 1000, 4, 4, 256, True, (-1, -1)),
]


@pytest.mark.parametrize("case", WIDE_FWD, ids=[str(c) for c in WIDE_FWD])
def test_hand_pipelined_forward_vs_oracle_and_general_kernel(case):
    """fwd_prefill_wide.hip (what d > 128 takes without ALiBi) against the oracle and against the
    general kernel (development library, switch 4 = 2): the two follow the running maximum with
    different slack, so they agree up to the rounding of P and O, and exactly in which rows see no
    key (LSE = +inf).  The output buffer is poisoned first: every row must be written."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT, f32, ulp_tol
    dt, b, sq, sk, h, hk, d, causal, window = case
    torch.manual_seed(3 * sq + sk)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    kw = dict(causal=causal, window=window)
    out, lse = cabi.fwd(q, k, v, out=torch.full_like(q, float("nan")), **kw)[:2]
    assert not torch.isnan(out).any()
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), dtype=dt, **kw)
    assert np.abs(f32(out) - ro).max() <= ulp_tol(dt, ro)
    got = lse.cpu().numpy()
    inf = np.isposinf(rl)
    assert (np.isposinf(got) == inf).all()
    assert np.abs(got[~inf] - rl[~inf]).max(initial=0.0) <= 1e-3
    with cabi.dev(4, 2):
        o0, l0 = cabi.fwd(q, k, v, **kw)[:2]
    assert np.abs(f32(out) - f32(o0)).max() <= 2 * ulp_tol(dt, ro)
    assert (np.isposinf(l0.cpu().numpy()) == inf).all()


def test_hand_pipelined_forward_kvcache_call_and_strided_inputs():
    """The kv-cache entry point at d = 256 with 200 query tokens per sequence (prefill against a
    cache: per-sequence key counts, a batch-index indirection into the cache, causal alignment at
    the bottom right), and q / k / v as strided views of one packed tensor."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(9)
    b, sq, cap, h, hk, d = 3, 200, 900, 4, 2, 256
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(b + 2, cap, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    lens = torch.tensor([900, 333, 200], dtype=torch.int32, device="cuda")
    idx = torch.tensor([4, 0, 2], dtype=torch.int32, device="cuda")
    out, _ = cabi.fwd(q, kc, vc, causal=True, cache_seqlens=lens, cache_batch_idx=idx, kvcache=True)[:2]
    ro, _ = fa_oracle.attention_fwd(f32(q), f32(kc[idx.long()]), f32(vc[idx.long()]), dtype="bf16", causal=True,
                                    cache_seqlens=lens.cpu().numpy())
    assert np.abs(f32(out) - ro).max() <= ulp_tol("bf16", ro)
    qkv = torch.randn(2, 600, 3, 2, d, device="cuda", dtype=torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out, lse = cabi.fwd(q, k, v, causal=True)[:2]
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), dtype="bf16", causal=True)
    assert np.abs(f32(out) - ro).max() <= ulp_tol("bf16", ro)
    assert np.abs(lse.cpu().numpy() - rl).max() <= 1e-3


@pytest.mark.parametrize("paged", [False, True])
@pytest.mark.parametrize("causal", [False, True])
def test_varlen_seqused_k(paged, causal):
    """seqused_k (flash_api.cpp:429,531-536; block_info.h:33): per sequence, only the first
    seqused_k[b] keys of the cu_seqlens_k extent are attended, also over a paged cache (the call
    vLLM makes: cu_seqlens_k spans the allocated pages, seqused_k the tokens in them)."""
    import cabi
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    import vllm_flash_attn_2_cuda as ext
    torch.manual_seed(7 + paged)
    dt, h, hk, d = "bf16", 4, 2, 128
    ql = [5, 64, 33]
    kl = [192, 256, 64]           # allocated extents
    used = [130, 256, 17]         # keys in use
    cq, ck = _cu(ql), _cu(kl)
    su = torch.tensor(used, dtype=torch.int32, device="cuda")
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    # oracle: the same batch with the keys cut to the used prefix
    kcut = torch.cat([k[ck[i]:ck[i] + used[i]] for i in range(3)])
    vcut = torch.cat([v[ck[i]:ck[i] + used[i]] for i in range(3)])
    cku = _cu(used)
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(kcut), f32(vcut), dtype=dt, causal=causal,
                                     cu_seqlens_q=cq.cpu().numpy(), cu_seqlens_k=cku.cpu().numpy(),
                                     max_seqlen_q=max(ql), max_seqlen_k=max(used))
    # and the oracle's own seqused_k path must agree with that
    ro2, _ = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), dtype=dt, causal=causal,
                                     cu_seqlens_q=cq.cpu().numpy(), cu_seqlens_k=ck.cpu().numpy(),
                                     max_seqlen_q=max(ql), max_seqlen_k=max(kl), seqused_k=np.array(used, np.int32))
    assert np.abs(ro - ro2).max() <= 1e-6
    if paged:
        page = 64
        nblk = sum(kl) // page
        perm = torch.randperm(nblk, device="cuda")
        kp = torch.empty(nblk, page, hk, d, device="cuda", dtype=torch.bfloat16)
        vp = torch.empty_like(kp)
        maxb = max(kl) // page
        bt = torch.zeros(3, maxb, dtype=torch.int32, device="cuda")
        nxt = 0
        for i in range(3):
            for j in range(kl[i] // page):
                pg = int(perm[nxt]); nxt += 1
                kp[pg] = k[ck[i] + j * page: ck[i] + (j + 1) * page]
                vp[pg] = v[ck[i] + j * page: ck[i] + (j + 1) * page]
                bt[i, j] = pg
        out, lse = cabi.fwd(q, kp, vp, causal=causal, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql),
                            max_seqlen_k=max(kl), seqused_k=su, block_table=bt)[:2]
        res = ext.varlen_fwd(q, kp, vp, None, cq, ck, su, bt, None, max(ql), max(kl), 0.0, d ** -0.5, False,
                             causal, -1, -1, False, None)
    else:
        out, lse = cabi.fwd(q, k, v, causal=causal, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql),
                            max_seqlen_k=max(kl), seqused_k=su)[:2]
        res = ext.varlen_fwd(q, k, v, None, cq, ck, su, None, None, max(ql), max(kl), 0.0, d ** -0.5, False,
                             causal, -1, -1, False, None)
    assert np.abs(f32(out) - ro).max() <= ulp_tol(dt, ro)
    for i in range(3):  # (causal with more query rows than keys in use: leading rows see nothing, LSE = +inf)
        a, r_ = f32(lse)[i, :, :ql[i]], rl[i, :, :ql[i]]
        assert np.array_equal(np.isposinf(a), np.isposinf(r_))
        fin = np.isfinite(r_)
        assert np.abs(a[fin] - r_[fin]).max() <= 1e-3
    assert torch.equal(res[0], out)  # the extension module's 19-argument entry plumbs seqused_k too


EIGHT_WAVE = [
    # head dims below the padded one at lengths that take the 8-wave kernels, whose prologues pass
    # the Q (and dO) rows through LDS images and the K fragments through buffer loads: the chunks
    # of the head-dim padding must read as zeros (Q, K) or be redirected to real ones (dQ kernel)
    ("bf16", 1, 1152, 1152, 2, 2, 96, False, (-1, -1), False),
    ("fp16", 1, 1152, 1152, 2, 1, 80, True, (-1, -1), False),
    ("bf16", 1, 1100, 1300, 2, 2, 72, True, (-1, -1), True),     # ragged last row block and key block
    ("bf16", 1, 600, 1100, 2, 2, 40, False, (300, 100), False),  # d = 64 kernels
    ("fp16", 1, 1030, 1030, 1, 1, 128, True, (-1, -1), False),
]


@pytest.mark.parametrize("case", EIGHT_WAVE, ids=[str(c) for c in EIGHT_WAVE])
def test_padded_head_dim_in_the_eight_wave_kernels(case):
    import cabi
    from oracle import fa_oracle
    from gpu_util import TDT, f32, ulp_tol
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    torch.manual_seed(d + sq)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    do = torch.randn_like(q)
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    kw = dict(causal=causal, window=window, alibi_slopes=slopes)
    okw = dict(dtype=dt, causal=causal, window=window, alibi_slopes=None if slopes is None else f32(slopes))
    out, lse = cabi.fwd(q, k, v, **kw)[:2]
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), **okw)
    assert np.abs(f32(out) - ro).max() <= ulp_tol(dt, ro)
    grads = cabi.bwd(do, q, k, v, out, lse, **kw)[:3]
    ref = fa_oracle.attention_bwd(f32(do), f32(q), f32(k), f32(v), ro, rl, **okw)[:3]
    _check_grads(grads, ref, dt)


# ---- head dims 65..96 and 129..192: the kernels stop their products at column 96 / 160 / 192 ----
CUT = [
    # b, sq, sk, h, hk, d, causal, window
    (2, 777, 777, 4, 4, 96, True, (-1, -1)),
    (2, 777, 1300, 4, 2, 80, False, (-1, -1)),        # grouped heads, key tail
    (1, 2100, 2100, 8, 8, 72, True, (-1, -1)),
    (2, 640, 900, 4, 1, 96, True, (300, 0)),          # sliding window: one block per workgroup
    (16, 4096, 4096, 32, 32, 96, True, (-1, -1)),     # persistent forward and dQ, 8-wave dK/dV
    (8, 4096, 4096, 32, 32, 96, False, (-1, -1)),     # full attention over long key ranges
    # head dims 129..192: the one-wave-per-SIMD kernels stop at column 160 or 192 of their 256
    (2, 777, 777, 4, 4, 160, True, (-1, -1)),
    (1, 1300, 2100, 4, 2, 144, False, (-1, -1)),
    (2, 1500, 1500, 4, 1, 192, True, (-1, -1)),
    (1, 2000, 2000, 2, 2, 176, True, (500, 0)),
    (4, 4096, 4096, 16, 16, 192, True, (-1, -1)),
    (2, 4096, 4096, 16, 4, 160, False, (-1, -1)),
]


@pytest.mark.parametrize("case", CUT, ids=[str(c) for c in CUT])
def test_cut_off_head_dims_equal_the_zero_padded_call_bit_for_bit(case):
    """A head dim of 65..96 runs the d = 128 tiles with the MFMA loops cut off at column 96 (129..192:
    the d = 256 tiles cut off at 160 or 192).  What is skipped multiplies zero padding, so the same
    tensors padded with zeros to d = 128 (256), which take the full loops and at these sizes partly
    other kernels, must give the same bits: output, LSE, softmax_d and all three gradients."""
    import cabi
    import torch.nn.functional as F
    b, sq, sk, h, hk, d, causal, window = case
    torch.manual_seed(d + sq)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q)
    kw = dict(causal=causal, window=window, softmax_scale=d ** -0.5)
    out, lse = cabi.fwd(q, k, v, **kw)
    dq, dk, dv, sd = cabi.bwd(do, q, k, v, out, lse, **kw)
    dp = 128 if d <= 128 else 256
    pad = lambda t: F.pad(t, (0, dp - d))
    out_p, lse_p = cabi.fwd(pad(q), pad(k), pad(v), **kw)
    dq_p, dk_p, dv_p, sd_p = cabi.bwd(pad(do), pad(q), pad(k), pad(v), out_p, lse_p, **kw)
    assert torch.equal(out, out_p[..., :d]) and not out_p[..., d:].any()
    assert torch.equal(lse, lse_p) or torch.equal(torch.nan_to_num(lse), torch.nan_to_num(lse_p))
    assert torch.equal(sd[:, :, :sq], sd_p[:, :, :sq])
    for a, a_p in ((dq, dq_p), (dk, dk_p), (dv, dv_p)):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, a_p[..., :d]) and not a_p[..., d:].any()


@pytest.mark.parametrize("d", [72, 96])
@pytest.mark.parametrize("splits", [2, 5])
def test_split_prefill_at_cut_off_head_dims_equals_the_zero_padded_call(d, splits):
    """The split-KV variants of the prefill kernel (chunked prefill against a long cache in a small
    batch) stop their products at column 96 too: same bits as the call padded to d = 128, and the
    oracle's result."""
    import cabi
    import torch.nn.functional as F
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(d + splits)
    b, sq, sk, h, hk = 1, 200, 3000, 4, 2
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    kw = dict(causal=True, softmax_scale=d ** -0.5, kvcache=True, num_splits=splits)
    out, lse = cabi.fwd(q, k, v, **kw)
    pad = lambda t: F.pad(t, (0, 128 - d))
    out_p, lse_p = cabi.fwd(pad(q), pad(k), pad(v), **kw)
    assert torch.equal(out, out_p[..., :d]) and torch.equal(lse, lse_p)
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(k), f32(v), dtype="bf16", causal=True)[:2]
    assert np.abs(f32(out) - ro).max() <= 2 * ulp_tol("bf16", ro)
    assert np.abs(lse.cpu().numpy() - rl).max() <= 2e-3
