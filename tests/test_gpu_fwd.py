"""GPU parity tests of the forward path, called through the C ABI (tests/cabi.py -> libvfa_hip.so)
and through the Python package (vllm_flash_attn -> vllm_flash_attn_2_cuda -> C ABI).

Checks: (1) the golden vectors (reference PyTorch oracle) with the reference tests' own rule
max|out - out_ref| <= 2 * max|out_pt - out_ref| + 1e-5 (tests/test_flash_attn.py:1087);
(2) the C oracle on seeded random inputs, to 2 ulp of the 16-bit output; (3) size-independent
properties at the benchmark configurations' full sizes."""
import numpy as np
import pytest
import torch

from util import bits_to_f32, expected_lse, golden_names, load_golden

pytestmark = pytest.mark.gpu


def _cabi():
    import cabi
    return cabi


def _cu(lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")


def _run_dense_golden(g, via):
    from gpu_util import to_dev
    m = g["meta"]
    dt = m["dtype"]
    q, k, v = (to_dev(g[x], dt) for x in "qkv")
    slopes = torch.from_numpy(g["alibi_slopes"]).cuda() if "alibi_slopes" in g else None
    kw = dict(causal=m["causal"], window=tuple(m["window"]), alibi_slopes=slopes)
    if m.get("qlens"):
        ql, kl = m["qlens"], m["klens"]
        qu = torch.cat([q[i, :n] for i, n in enumerate(ql)])
        ku = torch.cat([k[i, :n] for i, n in enumerate(kl)])
        vu = torch.cat([v[i, :n] for i, n in enumerate(kl)])
        cq, ck = _cu(ql), _cu(kl)
        if via == "cabi":
            out_u, lse = _cabi().fwd(qu, ku, vu, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql),
                                     max_seqlen_k=max(kl), **kw)
        else:
            import vllm_flash_attn as fa
            out_u, lse, _ = fa.flash_attn_varlen_func(qu, ku, vu, cq, ck, max(ql), max(kl), causal=m["causal"],
                                                      window_size=tuple(m["window"]), alibi_slopes=slopes,
                                                      return_attn_probs=True)
        out = torch.zeros_like(q)
        for i, n in enumerate(ql):
            out[i, :n] = out_u[cq[i]:cq[i + 1]]
        return out, lse
    if via == "cabi":
        return _cabi().fwd(q, k, v, **kw)
    import vllm_flash_attn as fa
    out, lse, _ = fa.flash_attn_func(q, k, v, causal=m["causal"], window_size=tuple(m["window"]),
                                     alibi_slopes=slopes, return_attn_probs=True)
    return out, lse


def _check_golden(g, out, lse, mult=2.0):
    m = g["meta"]
    out, lse = out.float().cpu().numpy(), lse.float().cpu().numpy()
    qlens = m.get("qlens") or [m["sq"]] * m["b"]
    tol = mult * float(g["err_out_pt"]) + 1e-5
    lse_ref = expected_lse(g)
    for i, n in enumerate(qlens):
        err = np.abs(out[i, :n] - g["out_ref"][i, :n]).max()
        assert err <= tol, (m["name"], i, err, tol)
        a, r = lse[i].reshape(m["h"], -1)[:, :n], lse_ref[i, :, :n]
        assert np.array_equal(np.isposinf(a), np.isposinf(r)), m["name"]
        fin = np.isfinite(r)
        if fin.any():
            assert np.abs(a[fin] - r[fin]).max() <= 2e-3 * max(1.0, np.abs(r[fin]).max()), m["name"]


DENSE_16 = [n for n in golden_names("dense") if "fp32" not in n]


@pytest.mark.parametrize("name", DENSE_16)
def test_fwd_cabi_vs_golden(name):
    g = load_golden(name)
    out, lse = _run_dense_golden(g, "cabi")
    _check_golden(g, out, lse)


@pytest.mark.parametrize("name", DENSE_16)
def test_fwd_python_api_vs_golden(name):
    g = load_golden(name)
    out, lse = _run_dense_golden(g, "py")
    _check_golden(g, out, lse)


@pytest.mark.parametrize("name", golden_names("kvcache"))
@pytest.mark.parametrize("num_splits", [1, 3, 0])
def test_kvcache_vs_golden(name, num_splits):
    from gpu_util import to_dev
    import vllm_flash_attn as fa
    g = load_golden(name)
    m = g["meta"]
    dt = m["dtype"]
    q = to_dev(g["q"], dt)
    cs = torch.from_numpy(g["cache_seqlens"]).cuda()
    slopes = torch.from_numpy(g["alibi_slopes"]).cuda() if "alibi_slopes" in g else None
    if m["page"]:
        kc, vc = to_dev(g["k_paged"], dt), to_dev(g["v_paged"], dt)
        bt = torch.from_numpy(g["block_table"]).cuda()
    else:
        kc, vc, bt = to_dev(g["k"], dt), to_dev(g["v"], dt), None
    out = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cs, block_table=bt, causal=m["causal"],
                                     window_size=tuple(m["window"]), alibi_slopes=slopes,
                                     num_splits=num_splits)
    mult = 5.0 if m["alibi"] else 3.0  # tests/test_flash_attn.py:2075-2076
    err = np.abs(out.float().cpu().numpy() - g["out_ref"]).max()
    assert err <= mult * float(g["err_out_pt"]) + 1e-5, (name, num_splits, err)
    # the same call through the C ABI, with the LSE
    if num_splits >= 1:
        out2, lse = _cabi().fwd(q, kc, vc, cache_seqlens=cs, block_table=bt, causal=m["causal"],
                                window=tuple(m["window"]), alibi_slopes=slopes, num_splits=num_splits)
        err2 = np.abs(out2.float().cpu().numpy() - g["out_ref"]).max()
        assert err2 <= mult * float(g["err_out_pt"]) + 1e-5
        lse_ref = expected_lse(g)
        fin = np.isfinite(lse_ref)
        assert np.abs(lse.cpu().numpy()[fin] - lse_ref[fin]).max() <= 2e-3 * max(1.0, np.abs(lse_ref[fin]).max())


CASES = [
    # dtype, b, sq, sk, h, hk, d, causal, window, alibi
    ("bf16", 2, 512, 512, 4, 4, 128, True, (-1, -1), False),
    ("bf16", 1, 1024, 1024, 2, 1, 128, False, (-1, -1), False),
    ("fp16", 2, 300, 777, 4, 2, 64, True, (-1, -1), False),
    ("bf16", 1, 640, 640, 4, 4, 64, True, (-1, -1), False),
    ("fp16", 1, 257, 513, 2, 2, 128, False, (100, 20), False),
    ("bf16", 2, 384, 384, 4, 2, 128, True, (128, 0), True),
    ("fp16", 1, 200, 333, 3, 3, 96, True, (-1, -1), False),
    ("bf16", 1, 130, 260, 2, 2, 256, False, (-1, -1), True),
    ("bf16", 1, 96, 160, 2, 2, 32, True, (-1, -1), False),
    ("fp16", 1, 16, 2000, 8, 2, 128, True, (-1, -1), False),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_fwd_vs_oracle_random(case):
    from oracle import fa_oracle
    from gpu_util import TDT, ulp_tol
    dt, b, sq, sk, h, hk, d, causal, window, alibi = case
    torch.manual_seed(0)
    q = torch.randn(b, sq, h, d, device="cuda", dtype=TDT[dt])
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=TDT[dt])
    slopes = (torch.rand(b, h, device="cuda") * 0.3) if alibi else None
    out, lse = _cabi().fwd(q, k, v, causal=causal, window=window, alibi_slopes=slopes)
    ro, rl = fa_oracle.attention_fwd(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(),
                                     dtype=dt, causal=causal, window=window,
                                     alibi_slopes=None if slopes is None else slopes.cpu().numpy())
    o = out.float().cpu().numpy()
    assert np.abs(o - ro).max() <= ulp_tol(dt, ro)
    l = lse.cpu().numpy()
    assert np.array_equal(np.isposinf(l), np.isposinf(rl))
    fin = np.isfinite(rl)
    assert np.abs(l[fin] - rl[fin]).max() <= 1e-3


def test_varlen_paged_vs_oracle():
    """flash_attn_varlen_func with a paged KV cache (block_table) against the oracle."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(1)
    b, h, hk, d, page = 3, 4, 2, 128, 16
    ql, kl = [70, 1, 129], [200, 333, 129]
    nblk_seq = [(n + page - 1) // page for n in kl]
    maxblk = max(nblk_seq)
    nblocks = b * maxblk + 5
    kc = torch.randn(nblocks, page, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(nblocks, page, hk, d, device="cuda", dtype=torch.bfloat16)
    perm = torch.randperm(nblocks)[: b * maxblk].reshape(b, maxblk).to(torch.int32).cuda()
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    cq, ck = _cu(ql), _cu(kl)
    out, lse = _cabi().fwd(q, kc, vc, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql),
                           max_seqlen_k=max(kl), block_table=perm, causal=True)
    ro, rl = fa_oracle.attention_fwd(q.float().cpu().numpy(), kc.float().cpu().numpy(), vc.float().cpu().numpy(),
                                     dtype="bf16", causal=True, cu_seqlens_q=cq.cpu().numpy(),
                                     cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=max(ql), max_seqlen_k=max(kl),
                                     block_table=perm.cpu().numpy())
    assert np.abs(out.float().cpu().numpy() - ro).max() <= ulp_tol("bf16", ro)
    import vllm_flash_attn as fa
    out2 = fa.flash_attn_varlen_func(q, kc, vc, cq, ck, max(ql), max(kl), causal=True, block_table=perm)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("d,page,sq", [(128, 16, 300), (128, 64, 700), (128, 256, 4200), (128, 32, 4200),
                                       (64, 32, 500), (64, 16, 500), (256, 16, 260), (128, 48, 300),
                                       (192, 16, 300), (96, 32, 300), (40, 16, 200), (256, 64, 700),
                                       (64, 128, 700), (224, 128, 4200), (160, 64, 700), (144, 16, 1300)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("staged", [False, True])
def test_varlen_paged_prefill_equals_contiguous(d, page, sq, causal, staged):
    """Prefill over a paged cache must give the bits of the same call on contiguous K/V: covers the
    per-page buffer-descriptor loads (power-of-two pages of at least 16 rows; straight into LDS, or
    -- development library, switch 5 = 1 -- through registers where a page holds whole load rounds) and the
    per-row path (small or odd pages), head dims below the padded one (whose padding chunks are
    redirected to real ones), ragged lengths that end inside a page, NaN in every cache row past a
    sequence's end and in unused pages, and garbage in unused block-table entries."""
    import vllm_flash_attn as fa
    import cabi
    if staged and (d > 128 or page % 16 or page & (page - 1)):
        pytest.skip("register staging is an alternative only for d <= 128 and power-of-two pages")
    if staged:
        def run(q, k, v, cq, ck, mq, mk, causal, block_table=None):
            return cabi.fwd(q, k, v, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=mq, max_seqlen_k=mk,
                            causal=causal, block_table=block_table)[0]
        with cabi.dev(5, 1):
            _paged_prefill_equals_contiguous(run, d, page, sq, causal)
    else:
        _paged_prefill_equals_contiguous(fa.flash_attn_varlen_func, d, page, sq, causal)


def _paged_prefill_equals_contiguous(varlen_func, d, page, sq, causal):
    torch.manual_seed(d + page)
    h, hk = 4, 2
    kl = [sq, sq - page // 2 - 3, 37]
    ql = [sq, 129, 37]
    b = len(kl)
    maxblk = (max(kl) + page - 1) // page + 2
    nblocks = b * maxblk + 3
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.bfloat16)
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.full((nblocks, page, hk, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    vc = torch.full_like(kc, float("nan"))
    perm = torch.randperm(nblocks)[: b * maxblk].reshape(b, maxblk).to(torch.int32)
    bt = perm.clone()
    off = 0
    for i, n in enumerate(kl):
        for j in range((n + page - 1) // page):
            rows = min(page, n - j * page)
            kc[perm[i, j], :rows] = k[off + j * page: off + j * page + rows]
            vc[perm[i, j], :rows] = v[off + j * page: off + j * page + rows]
        bt[i, (n + page - 1) // page:] = 2 ** 30  # never dereferenced
        off += n
    cq, ck = _cu(ql), _cu(kl)
    ref = varlen_func(q, k, v, cq, ck, max(ql), max(kl), causal=causal)
    out = varlen_func(q, kc, vc, cq, ck, max(ql), max(kl), causal=causal, block_table=bt.cuda())
    assert not torch.isnan(out).any()
    if page & (page - 1) == 0:
        assert torch.equal(out, ref)   # (d > 128: both through fwd_prefill_wide.hip, pages of 16 / 64 / 128 rows)
    else:
        # pages that are not a power of two take per-row addressing in the general kernel, while the
        # contiguous call of this short ragged batch packs its grouped heads into rows: two
        # instantiations, equal up to the rounding of P and O
        assert (out.float() - ref.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, ref.float().abs().max().item())


@pytest.mark.parametrize("interleaved", [True, False])
@pytest.mark.parametrize("paged", [False, True])
def test_kvcache_append_rotary_vs_oracle(interleaved, paged):
    """k/v append in place + rotary on K and Q + cache_batch_idx (tests/test_flash_attn.py:1815-2076
    exercises the same arguments; rotary has no reference-generated vector here: parity unpinned
    beyond the oracle's restatement of rotary.h)."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(2)
    b, sq, sk, h, hk, d, rd, snew = 3, 2, 256, 4, 2, 128, 64, 2
    dt = torch.float16
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dt)
    knew = torch.randn(b, snew, hk, d, device="cuda", dtype=dt)
    vnew = torch.randn(b, snew, hk, d, device="cuda", dtype=dt)
    cs = torch.tensor([5, 100, 254], dtype=torch.int32, device="cuda")
    ang = torch.rand(sk, rd // 2, device="cuda") * 6.28
    cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    if paged:
        page = 64
        nb = b * (sk // page) + 3
        kc = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
        vc = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
        bt = torch.randperm(nb)[: b * (sk // page)].reshape(b, -1).to(torch.int32).cuda()
        cbi = None
    else:
        kc = torch.randn(b + 2, sk, hk, d, device="cuda", dtype=dt)
        vc = torch.randn(b + 2, sk, hk, d, device="cuda", dtype=dt)
        bt = None
        cbi = torch.tensor([4, 0, 2], dtype=torch.int32, device="cuda")
    np_ = lambda t: None if t is None else (t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy())
    ro, rl, rk, rv = fa_oracle.attention_fwd(np_(q), np_(kc), np_(vc), dtype="fp16", causal=True,
                                             cache_seqlens=np_(cs), cache_batch_idx=np_(cbi), block_table=np_(bt),
                                             knew=np_(knew), vnew=np_(vnew), rotary_cos=np_(cos), rotary_sin=np_(sin),
                                             rotary_interleaved=interleaved)
    out = fa.flash_attn_with_kvcache(q, kc, vc, k=knew, v=vnew, rotary_cos=cos, rotary_sin=sin, cache_seqlens=cs,
                                     cache_batch_idx=cbi, block_table=bt, causal=True,
                                     rotary_interleaved=interleaved)
    assert np.array_equal(np_(vc), rv)                        # appended V: exact copy
    assert np.abs(np_(kc) - rk).max() <= 2.0 ** -10 * 8      # rotated K: 1 ulp of fp16 at |x| < 8
    assert np.abs(np_(out) - ro).max() <= 4 * ulp_tol("fp16", ro)


def test_repeatable_and_batch_independent():
    """Bitwise repeatability (tests/test_flash_attn.py:2135-2173 runs it 250x) and independence of
    batch entries: a batch slice computed alone is bitwise equal."""
    torch.manual_seed(3)
    q = torch.randn(4, 1024, 8, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(4, 1024, 2, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(4, 1024, 2, 128, device="cuda", dtype=torch.bfloat16)
    o0, l0 = _cabi().fwd(q, k, v, causal=True)
    for _ in range(250):
        o1, l1 = _cabi().fwd(q, k, v, causal=True)
        assert torch.equal(o0, o1) and torch.equal(l0, l1)
    o2, l2 = _cabi().fwd(q[1:2].contiguous(), k[1:2].contiguous(), v[1:2].contiguous(), causal=True)
    assert torch.equal(o0[1:2], o2) and torch.equal(l0[1:2], l2)


def test_strided_inputs_and_out_arg():
    """Arbitrary batch/row/head strides with unit last stride (flash_api.cpp:58-72): packed qkv views."""
    import vllm_flash_attn as fa
    torch.manual_seed(4)
    qkv = torch.randn(2, 300, 3, 4, 64, device="cuda", dtype=torch.float16)
    ref = fa.flash_attn_func(qkv[:, :, 0].contiguous(), qkv[:, :, 1].contiguous(), qkv[:, :, 2].contiguous(), causal=True)
    o1 = fa.flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True)
    o2 = fa.flash_attn_qkvpacked_func(qkv, causal=True)
    buf = torch.empty_like(ref)
    o3 = fa.flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True, out=buf)
    assert torch.equal(ref, o1) and torch.equal(ref, o2) and torch.equal(ref, o3) and o3.data_ptr() == buf.data_ptr()


def test_headline_shape_slices_vs_oracle():
    """Headline configuration (b8 s4096 h32 d128 bf16 causal), reduced to b=1 (the kernel treats
    batch entries independently, checked above): two (head) slices against the oracle."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(0)
    b, s, h, d = 1, 4096, 32, 128
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    out, lse = _cabi().fwd(q, k, v, causal=True)
    for hh in (0, 31):
        sl = lambda t: t[:, :, hh:hh + 1].float().cpu().numpy()
        ro, rl = fa_oracle.attention_fwd(sl(q), sl(k), sl(v), dtype="bf16", causal=True)
        assert np.abs(sl(out) - ro).max() <= ulp_tol("bf16", ro)
        assert np.abs(lse[:, hh:hh + 1].cpu().numpy() - rl).max() <= 1e-3
    # property: rows of softmax sum to one => with V = ones the output is exactly 1
    ones = torch.ones_like(v)
    o1, _ = _cabi().fwd(q, k, ones, causal=True)
    assert torch.equal(o1, torch.ones_like(o1))


def test_decode_config5_shape_properties():
    """Decode configuration per GPU shard (b128 sq1 kv8192 h32/hk8 d128 fp16), reduced batch:
    split vs unsplit agree, GQA folding equals the unfolded call, first batch entries vs oracle."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(0)
    b, sk, h, hk, d = 8, 8192, 32, 8, 128
    q = torch.randn(b, 1, h, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.float16)
    cs = torch.full((b,), sk, dtype=torch.int32, device="cuda")
    o1 = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cs, causal=True, num_splits=1)
    o4 = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cs, causal=True, num_splits=4)
    assert (o1.float() - o4.float()).abs().max().item() <= 2 * 2.0 ** -11
    o_unfolded, _ = _cabi().fwd(q, kc, vc, cache_seqlens=cs, causal=True)
    assert (o1.float() - o_unfolded.float()).abs().max().item() <= 2 * 2.0 ** -11
    f = lambda t: t[:2].float().cpu().numpy()
    ro, _ = fa_oracle.attention_fwd(f(q), f(kc), f(vc), dtype="fp16", cache_seqlens=cs[:2].cpu().numpy(), causal=True)
    assert np.abs(f(o1) - ro).max() <= ulp_tol("fp16", ro)


def test_decode_config5_full_batch():
    """The decode configuration at its full per-GPU size (b128 sq1 kv8192 h32/hk8 d128 fp16, 4.3 GB
    of K+V), checked through properties that do not need a CPU pass over the whole batch: a
    sequence computed alone equals its row of the batched call bit for bit, a paged cache in random
    page order equals the contiguous one bit for bit, ragged cache_seqlens rows equal the same rows
    computed alone, V = 1 gives 1, and the first / last / a ragged entry agree with the oracle."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(5)
    b, sk, h, hk, d = 128, 8192, 32, 8, 128
    q = torch.randn(b, 1, h, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.float16)
    full = torch.full((b,), sk, dtype=torch.int32, device="cuda")
    gen = torch.Generator(device="cpu").manual_seed(5)
    ragged = torch.randint(1, sk + 1, (b,), generator=gen, dtype=torch.int32)
    ragged[3], ragged[77] = 1, sk - 1
    ragged = ragged.cuda()

    def alone(i, lens):
        return fa.flash_attn_with_kvcache(q[i:i + 1], kc[i:i + 1], vc[i:i + 1], cache_seqlens=lens[i:i + 1],
                                          causal=True, num_splits=1)

    def check_oracle(o, i, lens):
        f = lambda t: t[i:i + 1].float().cpu().numpy()
        ro, _ = fa_oracle.attention_fwd(f(q), f(kc), f(vc), dtype="fp16",
                                        cache_seqlens=lens[i:i + 1].cpu().numpy(), causal=True)
        assert np.abs(f(o) - ro).max() <= ulp_tol("fp16", ro)

    o_full = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=full, causal=True, num_splits=1)
    assert torch.isfinite(o_full).all()
    for i in (0, 63, 127):
        assert torch.equal(o_full[i:i + 1], alone(i, full))
    check_oracle(o_full, 0, full)
    check_oracle(o_full, 127, full)
    # the library's own split count (what the reference's caller gets with num_splits = 0)
    o_auto = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=full, causal=True)
    assert (o_auto.float() - o_full.float()).abs().max().item() <= 2 * 2.0 ** -11

    o_rag = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=ragged, causal=True, num_splits=1)
    assert torch.isfinite(o_rag).all()
    for i in (3, 77, 100):
        assert torch.equal(o_rag[i:i + 1], alone(i, ragged))
    check_oracle(o_rag, 3, ragged)
    check_oracle(o_rag, 100, ragged)

    o_one = fa.flash_attn_with_kvcache(q, kc, torch.ones_like(vc), cache_seqlens=ragged, causal=True, num_splits=1)
    assert (o_one.float() - 1.0).abs().max().item() <= 2.0 ** -10
    del o_one

    page = 16
    nb = b * sk // page
    perm = torch.randperm(nb, device="cuda")
    bt = perm.view(b, sk // page).to(torch.int32)
    kp = torch.empty(nb, page, hk, d, device="cuda", dtype=torch.float16)
    vp = torch.empty_like(kp)
    kp[perm] = kc.view(nb, page, hk, d)
    vp[perm] = vc.view(nb, page, hk, d)
    o_paged = fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=ragged, block_table=bt, causal=True, num_splits=1)
    assert torch.equal(o_paged, o_rag)


def test_error_messages_match_reference():
    import vllm_flash_attn as fa
    q = torch.randn(1, 8, 4, 64, device="cuda", dtype=torch.float32)
    with pytest.raises(RuntimeError, match="only support fp16 and bf16"):
        fa.flash_attn_func(q, q, q)
    q = q.half()
    with pytest.raises(RuntimeError, match="p_dropout must be in"):  # flash_api.cpp:114
        fa.flash_attn_func(q, q, q, dropout_p=1.0)
    k = torch.randn(1, 8, 3, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="must divide number of heads in query"):
        fa.flash_attn_func(q, k, k)
    kc = torch.randn(4, 24, 4, 64, device="cuda", dtype=torch.float16)
    bt = torch.zeros(1, 2, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="divisible by 16"):
        fa.flash_attn_with_kvcache(q, kc, kc, block_table=bt, cache_seqlens=3)


def test_config2_shape_slices_vs_oracle():
    """Head-dim-64 forward configuration (b4 s2048 h16 d64 bf16 causal, fwd): head slices against the oracle."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(0)
    b, s, h, d = 4, 2048, 16, 64
    q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    out, lse = _cabi().fwd(q, k, v, causal=True)
    for bb, hh in ((0, 0), (3, 15)):
        sl = lambda t: t[bb:bb + 1, :, hh:hh + 1].float().cpu().numpy()
        ro, rl = fa_oracle.attention_fwd(sl(q), sl(k), sl(v), dtype="bf16", causal=True)
        assert np.abs(sl(out) - ro).max() <= ulp_tol("bf16", ro)
        assert np.abs(lse[bb:bb + 1, hh:hh + 1].cpu().numpy() - rl).max() <= 1e-3


def test_config4_varlen_gqa_window_alibi():
    """Ragged configuration: flash_attn_varlen_func, GQA 32/8, d128 bf16, ragged seqlens up to 8192,
    causal + window (1024, 0) + ALiBi.  Full batch on the GPU; parity vs the oracle on the short
    sequences (all heads) and on one head of the longest sequences; a sequence computed alone is
    bitwise equal to its slice of the batched call."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(0)
    lens = [8192, 6144, 4096, 3072, 2048, 1024, 513, 127]
    h, hk, d = 32, 8, 128
    total = sum(lens)
    cu = _cu(lens)
    q = torch.randn(total, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(total, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(total, hk, d, device="cuda", dtype=torch.bfloat16)
    slopes = torch.rand(len(lens), h, device="cuda") * 0.3
    out, lse, _ = fa.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True,
                                            window_size=(1024, 0), alibi_slopes=slopes, return_attn_probs=True)
    assert torch.isfinite(out.float()).all()
    cuh = cu.cpu().numpy()
    f = lambda t: t.float().cpu().numpy()
    # The oracle is called in ragged mode with the batch's max lengths so that it folds the window
    # like the batched call does (flash_api.cpp:515-516 uses max_seqlen_k: 1024 < 8192 stays a
    # local window even for the 513-token sequence, which selects the -slope*|i-j| ALiBi form and
    # with it the LSE convention).
    mx = max(lens)

    def oracle_seq(qs, ks, vs, sl, n):
        c = np.array([0, n], np.int32)
        return fa_oracle.attention_fwd(f(qs), f(ks), f(vs), dtype="bf16", causal=True, window=(1024, 0),
                                       alibi_slopes=f(sl), cu_seqlens_q=c, cu_seqlens_k=c,
                                       max_seqlen_q=mx, max_seqlen_k=mx)

    # short sequences, all heads (<= 1024 tokens in total per call)
    for i in (6, 7):
        a, e = cuh[i], cuh[i + 1]
        ro, rl = oracle_seq(q[a:e], k[a:e], v[a:e], slopes[i:i + 1], lens[i])
        assert np.abs(f(out[a:e]) - ro).max() <= ulp_tol("bf16", ro)
        assert np.abs(f(lse[i, :, :lens[i]]) - rl[0, :, :lens[i]]).max() <= 2e-3
        alone, lse1 = _cabi().fwd(q[a:e], k[a:e], v[a:e], cu_seqlens_q=_cu([lens[i]]), cu_seqlens_k=_cu([lens[i]]),
                                  max_seqlen_q=mx, max_seqlen_k=mx, causal=True, window=(1024, 0),
                                  alibi_slopes=slopes[i:i + 1].contiguous())
        assert torch.equal(alone, out[a:e])   # (both calls pack the grouped heads into rows: a windowed ragged batch)
    # one query head of the two longest sequences (true sliding window: 1024 < seqlen)
    for i, hq in ((0, 5), (1, 30)):
        a, e = cuh[i], cuh[i + 1]
        kvh = hq // (h // hk)
        ro, rl = oracle_seq(q[a:e, hq:hq + 1], k[a:e, kvh:kvh + 1], v[a:e, kvh:kvh + 1],
                            slopes[i:i + 1, hq:hq + 1], lens[i])
        assert np.abs(f(out[a:e, hq:hq + 1]) - ro).max() <= ulp_tol("bf16", ro)
        assert np.abs(f(lse[i, hq:hq + 1, :lens[i]]) - rl[0, :, :lens[i]]).max() <= 2e-3


def test_varlen_single_token_gqa_fold_and_empty_sequences():
    """varlen_fwd with max_seqlen_q == 1 and GQA takes the head-folding path of
    flash_api.cpp:500-507 (dense query, ragged keys); also a batch with an empty key sequence
    (output 0, LSE +inf for it) and an empty query sequence."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(5)
    b, h, hk, d = 4, 8, 2, 128
    kl = [300, 17, 1000, 64]
    q = torch.randn(b, h, d, device="cuda", dtype=torch.float16)
    k = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.float16)
    v = torch.randn(sum(kl), hk, d, device="cuda", dtype=torch.float16)
    cq, ck = _cu([1] * b), _cu(kl)
    out, lse, _ = fa.flash_attn_varlen_func(q, k, v, cq, ck, 1, max(kl), causal=True, return_attn_probs=True)
    f = lambda t: t.float().cpu().numpy()
    ro, rl = fa_oracle.attention_fwd(f(q), f(k), f(v), dtype="fp16", causal=True, cu_seqlens_q=cq.cpu().numpy(),
                                     cu_seqlens_k=ck.cpu().numpy(), max_seqlen_q=1, max_seqlen_k=max(kl))
    assert out.shape == q.shape and lse.shape == (b, h, 1)
    assert np.abs(f(out) - ro).max() <= ulp_tol("fp16", ro)
    assert np.abs(f(lse) - rl).max() <= 1e-3
    # ragged batch with an empty key sequence and an empty query sequence
    ql, kl = [5, 0, 40], [0, 30, 40]
    q = torch.randn(sum(ql), 2, 64, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(kl), 2, 64, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(sum(kl), 2, 64, device="cuda", dtype=torch.bfloat16)
    out, lse = _cabi().fwd(q, k, v, cu_seqlens_q=_cu(ql), cu_seqlens_k=_cu(kl), max_seqlen_q=40, max_seqlen_k=40)
    assert torch.equal(out[:5], torch.zeros_like(out[:5]))
    assert torch.isposinf(lse[0, :, :5]).all()
    ro, rl = fa_oracle.attention_fwd(f(q), f(k), f(v), dtype="bf16", cu_seqlens_q=_cu(ql).cpu().numpy(),
                                     cu_seqlens_k=_cu(kl).cpu().numpy(), max_seqlen_q=40, max_seqlen_k=40)
    assert np.abs(f(out) - ro).max() <= ulp_tol("bf16", ro)


def test_kvcache_without_seqlens_and_seqlen_k_zero():
    import vllm_flash_attn as fa
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(6)
    q = torch.randn(2, 3, 4, 64, device="cuda", dtype=torch.float16)
    kc = torch.randn(2, 100, 4, 64, device="cuda", dtype=torch.float16)
    vc = torch.randn(2, 100, 4, 64, device="cuda", dtype=torch.float16)
    out = fa.flash_attn_with_kvcache(q, kc, vc, causal=True)   # whole cache is valid
    f = lambda t: t.float().cpu().numpy()
    ro, _ = fa_oracle.attention_fwd(f(q), f(kc), f(vc), dtype="fp16", causal=True)
    assert np.abs(f(out) - ro).max() <= ulp_tol("fp16", ro)
    # seqlen_k == 0: flash_api.cpp:401-405
    k0 = torch.empty(2, 0, 4, 64, device="cuda", dtype=torch.float16)
    o0, l0, _ = fa.flash_attn_func(q, k0, k0, return_attn_probs=True)
    assert torch.equal(o0, torch.zeros_like(o0)) and torch.isposinf(l0).all()


@pytest.mark.parametrize("causal", [False, True])
def test_paged_kvcache_append_into_large_pool(causal):
    """The shapes of tests/test_flash_attn.py:2466-2507 (170 new tokens appended to an empty paged
    cache, 1000 pages of 16 rows, 32 heads, d 128), run 10 times like the reference.  There the
    page table is drawn with repetitions and the test only has to survive; here every other round
    draws distinct pages, so that the result can be checked: with empty caches the output is the
    attention of q over the new k / v alone, and the pages hold exactly the new rows."""
    import vllm_flash_attn as fa
    torch.manual_seed(21)
    b, sq, h, d, page, nblk = 4, 170, 32, 128, 16, 1000
    kc = torch.rand(nblk, page, h, d, device="cuda", dtype=torch.float16)
    vc = torch.rand_like(kc)
    lens = torch.zeros(b, dtype=torch.int32, device="cuda")
    per = (sq + page - 1) // page
    for it in range(10):
        q = torch.rand(b, sq, h, d, device="cuda", dtype=torch.float16)
        k = torch.rand_like(q)
        v = torch.rand_like(q)
        if it % 2:
            bt = torch.randint(0, nblk, (b, per), dtype=torch.int32, device="cuda")
        else:
            bt = torch.randperm(nblk, device="cuda")[:b * per].to(torch.int32).view(b, per)
        out = fa.flash_attn_with_kvcache(q, kc, vc, k=k, v=v, cache_seqlens=lens, block_table=bt, causal=causal)
        assert torch.isfinite(out).all()
        if it % 2:
            continue
        ref = fa.flash_attn_func(q, k, v, causal=causal)
        assert (out.float() - ref.float()).abs().max().item() <= 2 * 2.0 ** -11
        got_k = kc[bt.long()].reshape(b, per * page, h, d)[:, :sq]
        got_v = vc[bt.long()].reshape(b, per * page, h, d)[:, :sq]
        assert torch.equal(got_k, k) and torch.equal(got_v, v)


@pytest.mark.parametrize("num_splits", [0, 1, 3])
@pytest.mark.parametrize("paged", [False, True])
def test_kvcache_rotary_with_many_query_rows(num_splits, paged):
    """130 query rows with appended keys and rotary tables, unsplit and split: the split variants
    of the prefill kernel do not rotate the query, so these calls must stay on the kernel that
    does (a forced split count used to take them and return unrotated-query results)."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(12)
    b, sq, sk, h, hk, d, rd, page = 2, 130, 512, 8, 2, 128, 64, 32
    dt = torch.bfloat16
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dt)
    knew = torch.randn(b, sq, hk, d, device="cuda", dtype=dt)
    vnew = torch.randn_like(knew)
    cs = torch.tensor([300, 7], dtype=torch.int32, device="cuda")
    ang = torch.rand(sk, rd // 2, device="cuda") * 6.28
    cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    if paged:
        nb = b * (sk // page) + 2
        kc = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
        vc = torch.randn_like(kc)
        bt = torch.randperm(nb)[: b * (sk // page)].reshape(b, -1).to(torch.int32).cuda()
    else:
        kc = torch.randn(b, sk, hk, d, device="cuda", dtype=dt)
        vc = torch.randn_like(kc)
        bt = None
    np_ = lambda t: None if t is None else (t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy())
    ro = fa_oracle.attention_fwd(np_(q), np_(kc), np_(vc), dtype="bf16", causal=True, cache_seqlens=np_(cs),
                                 block_table=np_(bt), knew=np_(knew), vnew=np_(vnew), rotary_cos=np_(cos),
                                 rotary_sin=np_(sin), rotary_interleaved=True)[0]
    out = fa.flash_attn_with_kvcache(q, kc, vc, k=knew, v=vnew, rotary_cos=cos, rotary_sin=sin, cache_seqlens=cs,
                                     block_table=bt, causal=True, rotary_interleaved=True, num_splits=num_splits)
    assert np.abs(np_(out) - ro).max() <= 4 * ulp_tol("bf16", ro)


def test_kvcache_rotary_position_past_the_table_is_clamped():
    """More query rows than appended keys under a causal mask: the query's rotary position
    cache_seqlen + row runs past seqlen_k, where a cos / sin table of exactly seqlen_k rows ends
    (the reference kernel reads past it; its tests keep cache_seqlens low enough,
    tests/test_flash_attn.py:1955-1964).  Here the position stops at the table's last row, in the
    kernel and in the oracle alike, and the result does not depend on what follows the table."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    import vllm_flash_attn as fa
    torch.manual_seed(8)
    b, sq, sk, h, hk, d, rd = 2, 64, 128, 4, 2, 128, 64
    dt = torch.float16
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dt)
    knew = torch.randn(b, 1, hk, d, device="cuda", dtype=dt)
    vnew = torch.randn_like(knew)
    cs = torch.tensor([127, 100], dtype=torch.int32, device="cuda")
    big = torch.rand(sk + 64, rd // 2, device="cuda") * 6.28
    outs = []
    for fill in (0.0, float("nan")):
        store_c, store_s = torch.cos(big).to(dt), torch.sin(big).to(dt)
        store_c[sk:] = fill   # what lies behind the table must not matter
        store_s[sk:] = fill
        cos, sin = store_c[:sk], store_s[:sk]
        kc = torch.randn(b, sk, hk, d, device="cuda", dtype=dt, generator=torch.Generator("cuda").manual_seed(1))
        vc = torch.randn(b, sk, hk, d, device="cuda", dtype=dt, generator=torch.Generator("cuda").manual_seed(2))
        np_ = lambda t: t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy()
        ro = fa_oracle.attention_fwd(np_(q), np_(kc), np_(vc), dtype="fp16", causal=True, cache_seqlens=np_(cs),
                                     knew=np_(knew), vnew=np_(vnew), rotary_cos=np_(cos), rotary_sin=np_(sin),
                                     rotary_interleaved=False)[0]
        out = fa.flash_attn_with_kvcache(q, kc, vc, k=knew, v=vnew, rotary_cos=cos, rotary_sin=sin,
                                         cache_seqlens=cs, causal=True, rotary_interleaved=False)
        assert torch.isfinite(out).all()
        assert np.abs(np_(out) - ro).max() <= 4 * ulp_tol("fp16", ro)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def test_kvcache_gqa_decode_writes_caller_shaped_out():
    """seqlen_q = 1 with grouped heads takes the (b, ngroups, h_k, d) fold; an `out` of the caller's
    own shape (b, 1, h, d) is filled in place.  (The reference checks `out` against the folded shape
    there, flash_api.cpp:787, and raises; both shapes are accepted here.)"""
    import vllm_flash_attn as fa
    torch.manual_seed(11)
    b, kv, h, hk, d = 3, 300, 8, 2, 64
    q = torch.randn(b, 1, h, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    lens = torch.tensor([300, 17, 129], dtype=torch.int32, device="cuda")
    ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True)
    out = torch.full_like(q, float("nan"))
    got = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, out=out)
    assert got.shape == q.shape and torch.equal(out, ref) and torch.equal(got, ref)
    folded = torch.full((b, h // hk, hk, d), float("nan"), device="cuda", dtype=torch.float16)
    fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, out=folded)
    assert torch.equal(folded.transpose(1, 2).reshape(b, 1, h, d), ref)


def test_decode_is_graph_capturable():
    """vLLM replays its decode step from a captured graph: the whole call (wrapper, extension,
    C ABI) must only enqueue work on the capturing stream.  Capture the paged split-KV decode and
    the contiguous single-pass decode, change the inputs in place, replay, compare with eager."""
    import vllm_flash_attn as fa
    torch.manual_seed(12)
    b, kv, h, hk, d, page = 4, 1024, 8, 2, 128, 16
    dt = torch.float16
    q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=dt)
    vc = torch.randn_like(kc)
    lens = torch.tensor([1024, 700, 33, 512], dtype=torch.int32, device="cuda")
    nb = b * kv // page
    perm = torch.randperm(nb, device="cuda")
    bt = perm.view(b, kv // page).to(torch.int32)
    kp = torch.empty(nb, page, hk, d, device="cuda", dtype=dt)
    vp = torch.empty_like(kp)
    kp[perm] = kc.view(nb, page, hk, d)
    vp[perm] = vc.view(nb, page, hk, d)
    out_p = torch.empty_like(q)
    out_c = torch.empty_like(q)

    def step():
        fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True, out=out_p)
        fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, num_splits=1, out=out_c)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()  # warm-up outside capture (lazy module / attribute set-up)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    q.copy_(torch.randn_like(q))          # new query, same buffers
    lens.copy_(torch.tensor([5, 1024, 900, 64], dtype=torch.int32, device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    got_p, got_c = out_p.clone(), out_c.clone()
    ref = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, num_splits=1)
    assert torch.equal(got_c, ref)
    assert (got_p.float() - ref.float()).abs().max().item() <= 2e-3


@pytest.mark.parametrize("d,sq,sk,splits,causal,paged", [
    (128, 200, 1000, 2, True, False), (128, 130, 4100, 5, False, False), (64, 513, 2000, 3, True, False),
    (256, 100, 700, 4, True, False), (128, 300, 1500, 3, True, True), (64, 90, 900, 7, False, True),
    (128, 65, 257, 4, True, False)])
def test_split_kv_with_many_query_rows(d, sq, sk, splits, causal, paged):
    """num_splits > 1 with more than 64 query rows runs the prefill kernel's split variants +
    the combine kernel (a few hundred new tokens against a long context): against the oracle's
    split + combine and against the unsplit call; ragged second sequence, splits that see no key
    (causal, short sequence), GQA, paged cache."""
    from oracle import fa_oracle
    from gpu_util import f32, ulp_tol
    torch.manual_seed(sq + sk)
    b, h, hk, page = 2, 4, 2, 32
    lens = torch.tensor([sk, max(1, sk // 3)], dtype=torch.int32, device="cuda")
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    kw = dict(causal=causal, cache_seqlens=lens)
    okw = dict(dtype="bf16", causal=causal, cache_seqlens=lens.cpu().numpy())
    kk, vv = k, v
    if paged:
        nblk = (sk + page - 1) // page
        perm = torch.randperm(b * nblk, device="cuda").reshape(b, nblk).to(torch.int32)
        kk = torch.zeros(b * nblk, page, hk, d, device="cuda", dtype=torch.bfloat16)
        vv = torch.zeros_like(kk)
        pad = nblk * page - sk
        kk[perm.flatten().long()] = torch.nn.functional.pad(k, (0, 0, 0, 0, 0, pad)).reshape(b * nblk, page, hk, d)
        vv[perm.flatten().long()] = torch.nn.functional.pad(v, (0, 0, 0, 0, 0, pad)).reshape(b * nblk, page, hk, d)
        kw["block_table"] = perm
        okw["block_table"] = perm.cpu().numpy()
    out1, lse1 = _cabi().fwd(q, kk, vv, num_splits=1, **kw)
    outs, lses = _cabi().fwd(q, kk, vv, num_splits=splits, **kw)
    ro, rl = fa_oracle.attention_fwd(f32(q), f32(kk), f32(vv), num_splits=splits, block_n=64, block_m=128, **okw)
    assert np.abs(f32(outs) - ro).max() <= ulp_tol("bf16", ro)
    assert np.abs(f32(outs) - f32(out1)).max() <= ulp_tol("bf16", ro)
    fin = np.isfinite(rl)
    assert np.array_equal(np.isposinf(f32(lses)), np.isposinf(rl))
    assert np.abs(f32(lses)[fin] - rl[fin]).max() <= 1e-3
    assert np.abs(f32(lses)[fin] - f32(lse1)[fin]).max() <= 1e-3
