"""The forward kernel with 64 query rows per wave (one wave per SIMD, software-pipelined tile loop)
against the oracle and the 8-wave kernel, on shapes around its tile edges (256-row blocks, 64-key
tiles), with the masks it takes (causal, window, key tail) and grouped heads."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cabi():
    import cabi
    return cabi


def _with_kernel(fn):
    with _cabi().dev(4, 1):   # the kernel exists in the development library only
        return fn()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sq,sk,causal,window", [
    (256, 256, False, (-1, -1)), (256, 256, True, (-1, -1)), (512, 512, True, (-1, -1)),
    (300, 300, True, (-1, -1)), (64, 1000, True, (-1, -1)), (1000, 64, True, (-1, -1)),
    (513, 777, False, (-1, -1)), (700, 700, False, (100, 30)), (640, 900, False, (50, 0)),
    (1, 129, True, (-1, -1)), (257, 63, False, (-1, -1)), (1024, 1024, True, (-1, -1))])
@pytest.mark.parametrize("d,h,hk", [(128, 4, 4), (96, 6, 2), (72, 2, 1)])
def test_w64_forward_matches_oracle(dtype, sq, sk, causal, window, d, h, hk):
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(sq * 7 + sk)
    b = 2
    q = torch.randn(b, sq, h, d, device="cuda", dtype=dtype)
    k = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    v = torch.randn(b, sk, hk, d, device="cuda", dtype=dtype)
    out, lse = _with_kernel(lambda: _cabi().fwd(q, k, v, causal=causal, window=window))
    f = lambda t: t.float().cpu().numpy()
    name = "bf16" if dtype == torch.bfloat16 else "fp16"
    ro, rl = fa_oracle.attention_fwd(f(q), f(k), f(v), dtype=name, causal=causal, window=window)
    assert np.abs(f(out) - ro).max() <= ulp_tol(name, ro)
    got = lse.cpu().numpy()
    inf = np.isposinf(rl)
    assert (np.isposinf(got) == inf).all()
    assert np.abs(got[~inf] - rl[~inf]).max(initial=0.0) <= 1e-3


def test_w64_forward_varlen_and_large_scores():
    """Ragged batch through the varlen entry, and rows whose maximum moves late (the lazy reference
    point must rescale O that is already accumulated)."""
    from oracle import fa_oracle
    from gpu_util import ulp_tol
    torch.manual_seed(1)
    h, d = 4, 128
    lens = [300, 1, 64, 257, 900]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens), h, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(sum(lens), h, d, device="cuda", dtype=torch.bfloat16)
    out, lse = _with_kernel(lambda: _cabi().fwd(q, k, v, causal=True, cu_seqlens_q=cu, cu_seqlens_k=cu,
                                                max_seqlen_q=max(lens), max_seqlen_k=max(lens)))
    f = lambda t: t.float().cpu().numpy()
    ro, _ = fa_oracle.attention_fwd(f(q), f(k), f(v), dtype="bf16", causal=True, cu_seqlens_q=cu.cpu().numpy(),
                                    cu_seqlens_k=cu.cpu().numpy(), max_seqlen_q=max(lens), max_seqlen_k=max(lens))
    assert np.abs(f(out) - ro).max() <= ulp_tol("bf16", ro)

    # the tiles are walked from the last keys to the first: make the first keys the largest scores
    b, s = 1, 1024
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k[:, :64] *= 6.0
    k[:, 300:364] *= 3.0
    v = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    out, lse = _with_kernel(lambda: _cabi().fwd(q, k, v, causal=False))
    ro, rl = fa_oracle.attention_fwd(f(q), f(k), f(v), dtype="bf16", causal=False)
    assert np.abs(f(out) - ro).max() <= ulp_tol("bf16", ro)
    assert np.abs(lse.cpu().numpy() - rl).max() <= 2e-3
