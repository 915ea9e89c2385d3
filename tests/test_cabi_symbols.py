"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/vfa.h
declares; host-side helpers match the oracle; the Python package and extension import."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "vfa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vfa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import cabi
    lib = ctypes.CDLL(cabi.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vfa.h but not exported"


def test_product_library_exports_nothing_else():
    """The shipped library's dynamic symbol table is include/vfa.h: no development hook (the
    kernel-variant switches live in libvfa_hip_dev.so only) and no internal C++ symbol."""
    import subprocess
    import cabi
    out = subprocess.run(["nm", "-D", "--defined-only", cabi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    funcs = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TtWw")
    assert funcs == _declared_functions(), funcs
    dev = subprocess.run(["nm", "-D", "--defined-only", cabi.DEV_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "vfa_debug_set_tuning" in dev


def test_params_struct_mirror_matches():
    import cabi
    assert cabi.lib().vfa_params_sizeof() == ctypes.sizeof(cabi.VfaParams)


def test_split_heuristic_matches_oracle():
    import cabi
    from oracle import fa_oracle
    lib = cabi.lib()
    for bnm in (1, 7, 48, 100, 409, 410, 900):
        for nsm in (108, 512):
            for nblk in (1, 3, 64, 1000):
                assert lib.vfa_num_splits_heuristic(bnm, nsm, nblk, 128) == \
                    fa_oracle.num_splits_heuristic(bnm, nsm, nblk, 128)


def test_argument_checks_without_gpu():
    """Checks that fail before any HIP call return the reference's message."""
    import cabi
    P = cabi.VfaParams()
    P.dtype = 7
    rc = cabi.lib().vfa_mha_fwd(ctypes.byref(P))
    assert rc == -1 and b"only support fp16 and bf16" in cabi.lib().vfa_last_error()
    P.dtype = 1
    P.q = P.k = P.v = P.out = P.softmax_lse = 16
    P.b, P.h, P.h_k, P.d = 1, 6, 4, 64
    rc = cabi.lib().vfa_mha_fwd(ctypes.byref(P))
    assert rc == -1 and b"must divide number of heads in query" in cabi.lib().vfa_last_error()
    P.h_k = 3
    P.p_dropout = 1.0  # flash_api.cpp:114
    rc = cabi.lib().vfa_mha_fwd(ctypes.byref(P))
    assert rc == -1 and b"p_dropout must be in [0, 1)" in cabi.lib().vfa_last_error()
    P.p_dropout = 0.0
    P.s_dmask = 8  # flash_api.cpp:352
    rc = cabi.lib().vfa_mha_fwd(ctypes.byref(P))
    assert rc == -1 and b"return_softmax is only supported when p_dropout > 0.0" in cabi.lib().vfa_last_error()


def test_python_package_imports_and_signatures():
    import inspect
    import vllm_flash_attn as fa
    import vllm_flash_attn_2_cuda as ext
    assert fa.__version__ == "2.5.8.post3"
    for n in ("fwd", "varlen_fwd", "fwd_kvcache", "bwd", "varlen_bwd"):
        assert hasattr(ext, n)
    sig = inspect.signature(fa.flash_attn_with_kvcache)
    assert list(sig.parameters) == ["q", "k_cache", "v_cache", "k", "v", "rotary_cos", "rotary_sin",
                                    "cache_seqlens", "cache_batch_idx", "block_table", "softmax_scale",
                                    "causal", "window_size", "rotary_interleaved", "alibi_slopes",
                                    "num_splits", "out"]
    sig = inspect.signature(fa.flash_attn_varlen_func)
    assert list(sig.parameters) == ["q", "k", "v", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q",
                                    "max_seqlen_k", "dropout_p", "softmax_scale", "causal", "window_size",
                                    "alibi_slopes", "deterministic", "return_attn_probs", "block_table", "out"]
    sig = inspect.signature(fa.flash_attn_func)
    assert list(sig.parameters) == ["q", "k", "v", "dropout_p", "softmax_scale", "causal", "window_size",
                                    "alibi_slopes", "deterministic", "return_attn_probs", "out"]
    from vllm_flash_attn.flash_attn_interface import _get_block_size_n  # noqa: F401


def test_cpu_tensors_are_rejected_not_silently_computed():
    import torch
    import vllm_flash_attn as fa
    q = torch.randn(1, 4, 2, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fa.flash_attn_func(q, q, q)


def test_config1_fp32_cpu_through_the_api_matches_torch_sdpa():
    """BASELINE config 1: flash_attn_func, batch 2, seqlen 128, 4 heads, headdim 64, fp32,
    non-causal, on the CPU, against torch's scaled_dot_product_attention to 1e-5 -- through the
    package's own API (its host fp32 path; the oracle is not involved).  Plus the causal and GQA
    forms of the same plumbing, and LSE against logsumexp of the scaled scores."""
    import torch
    import torch.nn.functional as F
    import vllm_flash_attn as fa
    torch.manual_seed(0)
    b, s, h, d = 2, 128, 4, 64
    q, k, v = (torch.randn(b, s, h, d, dtype=torch.float32) for _ in range(3))
    # opt-in: without it fp32 CPU tensors get the reference's error, like any non-CUDA input
    with pytest.raises(RuntimeError, match="fp16 and bf16|must be on CUDA"):
        fa.flash_attn_func(q, k, v)
    prev = fa.enable_cpu_fp32(True)
    try:
        _config1_checks(fa, F, torch, q, k, v, b, s, h, d)
        with pytest.raises(RuntimeError, match="forward only"):
            fa.flash_attn_func(q.clone().requires_grad_(), k, v)
    finally:
        fa.enable_cpu_fp32(prev)


def _config1_checks(fa, F, torch, q, k, v, b, s, h, d):
    out, lse, sdm = fa.flash_attn_func(q, k, v, return_attn_probs=True)
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    assert out.dtype == torch.float32 and out.shape == q.shape
    assert (out - ref).abs().max().item() <= 1e-5
    scores = torch.einsum("bqhd,bkhd->bhqk", q, k) * d ** -0.5
    assert (lse - torch.logsumexp(scores, dim=-1)).abs().max().item() <= 1e-4
    assert sdm.numel() == 0
    out_c = fa.flash_attn_func(q, k, v, causal=True)
    ref_c = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                           is_causal=True).transpose(1, 2)
    assert (out_c - ref_c).abs().max().item() <= 1e-5
    kg, vg = k[:, :, :2].contiguous(), v[:, :, :2].contiguous()
    out_g = fa.flash_attn_func(q, kg, vg)
    ref_g = F.scaled_dot_product_attention(q.transpose(1, 2), kg.repeat_interleave(2, dim=2).transpose(1, 2),
                                           vg.repeat_interleave(2, dim=2).transpose(1, 2)).transpose(1, 2)
    assert (out_g - ref_g).abs().max().item() <= 1e-5
    # bottom-right aligned causal mask with more keys than queries, and a sliding window
    q2 = q[:, :40]
    out_w = fa.flash_attn_func(q2, k, v, window_size=(17, 5))
    i = torch.arange(40)[:, None] + (s - 40)
    j = torch.arange(s)[None, :]
    mask = (j <= i + 5) & (j >= i - 17)
    ref_w = F.scaled_dot_product_attention(q2.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                           attn_mask=mask).transpose(1, 2)
    assert (out_w - ref_w).abs().max().item() <= 1e-5


def test_decode_split_count_counts_packed_gqa_rows_like_the_launch():
    """Grouped heads, few query tokens: the planner's work items are (batch, KV head, 32-row block of
    seqlen_q * group rows), the decode kernel's grid, while the packed rows fit one such block, and
    the packed prefill kernel's 128-row blocks beyond; MHA and long queries keep the reference's count."""
    import cabi
    lib = cabi.lib()
    cus = 256
    for sq in (1, 4, 8, 9, 16, 33, 64):
        for h, hk in ((32, 8), (16, 2), (8, 1)):
            g = h // hk
            if sq * g > 32:
                # more than one decode block: the packed prefill kernel, 128-row blocks per KV head,
                # at least four 64-key tiles per split, nothing split once a quarter of the slots is filled
                blocks = 2 * hk * ((sq * g + 127) // 128)
                want = 1 if blocks * 4 > 2 * cus else lib.vfa_num_splits_heuristic(blocks, 2 * cus, (8192 // 64) // 4, 128)
            else:
                mb = (sq * g + 31) // 32
                want = min((8192 + 127) // 128, lib.vfa_num_splits_heuristic(2 * hk * mb, 2 * cus, (8192 + 127) // 128, 128))
            assert lib.vfa_decode_num_splits(2, h, hk, 128, 8192, sq, cus) == want, (sq, h, hk)
    assert lib.vfa_decode_num_splits(2, 8, 8, 128, 8192, 1, cus) == lib.vfa_splitkv_num_splits(2, 8, 128, 8192, 1, cus)
    assert lib.vfa_decode_num_splits(2, 32, 8, 128, 8192, 200, cus) == lib.vfa_splitkv_num_splits(2, 32, 128, 8192, 200, cus)
    # the case the old planner under-split: 16 tokens x 4 grouped heads = 64 rows = 2 blocks per KV head
    assert lib.vfa_decode_num_splits(1, 32, 8, 128, 8192, 16, cus) > lib.vfa_splitkv_num_splits(1, 32, 128, 8192, 16, cus)


def test_decode_split_count_never_exceeds_the_kernels_key_tiles():
    import cabi
    lib = cabi.lib()
    for d in (64, 128, 192, 256):
        for sk in (100, 300, 1000, 8192):
            n = lib.vfa_splitkv_num_splits(1, 2, d, sk, 1, 256)
            assert 1 <= n <= max(1, (sk + 127) // 128), (d, sk, n)
    # up to d = 128 the count is the reference's function of the shapes
    from oracle import fa_oracle
    assert lib.vfa_splitkv_num_splits(2, 8, 128, 8192, 1, 256) == fa_oracle.num_splits_heuristic(16, 512, 64, 128)
