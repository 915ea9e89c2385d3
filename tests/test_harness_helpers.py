"""The padding / rotary helpers shipped beside the attention functions (the `bert_padding` and
`layers.rotary` names the reference's tests import, tests/test_flash_attn.py:16-18): CPU checks of
the helpers themselves, and GPU checks that follow the reference's kv-cache test
(tests/test_flash_attn.py:1957-2076) and its ragged-batch set-up (:70-175) with them."""
import math

import numpy as np
import pytest
import torch

from oracle import fa_oracle
from vllm_flash_attn.bert_padding import pad_input, unpad_input
from vllm_flash_attn.layers.rotary import apply_rotary_emb


def test_unpad_pad_round_trip():
    torch.manual_seed(0)
    x = torch.randn(3, 9, 2, 8, requires_grad=True)
    lens = torch.tensor([9, 4, 1])
    mask = torch.arange(9)[None, :] < lens[:, None]
    u, idx, cu, mx = unpad_input(x, mask)
    assert u.shape == (14, 2, 8) and cu.tolist() == [0, 9, 13, 14] and cu.dtype == torch.int32 and mx == 9
    p = pad_input(u, idx, 3, 9)
    assert torch.equal(p, x * mask[:, :, None, None])
    p.sum().backward()
    assert torch.equal(x.grad, mask[:, :, None, None].expand_as(x).float())


@pytest.mark.parametrize("interleaved", [True, False])
@pytest.mark.parametrize("rotary_frac", [0.5, 1.0])
def test_rotary_helper_matches_oracle_append(interleaved, rotary_frac):
    """apply_rotary_emb(k_new, seqlen_offsets=cache_seqlens) is what the oracle's restatement of
    rotary.h writes into the cache (bf16 rounding of the fp32 rotation in both)."""
    rng = np.random.default_rng(1)
    b, sk, snew, hk, d = 2, 40, 3, 2, 64
    ro = int(d * rotary_frac)
    r16 = lambda a: fa_oracle.round_to(a.astype(np.float32), "bf16")
    kc, vc = r16(rng.standard_normal((b, sk, hk, d))), r16(rng.standard_normal((b, sk, hk, d)))
    kn, vn = r16(rng.standard_normal((b, snew, hk, d))), r16(rng.standard_normal((b, snew, hk, d)))
    q = r16(rng.standard_normal((b, snew, 4, d)))
    ang = rng.random((sk, ro // 2)) * 2 * math.pi
    cos, sin = r16(np.cos(ang)), r16(np.sin(ang))
    cache_seqlens = np.array([5, 30], dtype=np.int32)
    _, _, k_upd, v_upd = fa_oracle.attention_fwd(q, kc, vc, dtype="bf16", causal=True, cache_seqlens=cache_seqlens,
                                                 knew=kn, vnew=vn, rotary_cos=cos, rotary_sin=sin,
                                                 rotary_interleaved=interleaved)
    k_ro = apply_rotary_emb(torch.from_numpy(kn).bfloat16(), torch.from_numpy(cos).bfloat16(),
                            torch.from_numpy(sin).bfloat16(), seqlen_offsets=torch.from_numpy(cache_seqlens),
                            interleaved=interleaved).float().numpy()
    for i, c in enumerate(cache_seqlens):
        assert np.abs(k_upd[i, c:c + snew] - k_ro[i]).max() <= 2.0 ** -7 * max(1.0, np.abs(k_ro[i]).max())
        assert np.array_equal(v_upd[i, c:c + snew], vn[i])


def _attn_ref(q, k, v, key_mask, causal):
    """fp32 softmax attention with a key padding mask and the bottom-right causal mask."""
    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    k = k.float().repeat_interleave(h // hk, dim=2)
    v = v.float().repeat_interleave(h // hk, dim=2)
    s = torch.einsum("bthd,bshd->bhts", q.float() / math.sqrt(d), k)
    s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    if causal:
        klen = key_mask.sum(-1)
        row = torch.arange(sq, device=q.device)[None, :, None]
        col = torch.arange(sk, device=q.device)[None, None, :]
        s = s.masked_fill((col > row + klen[:, None, None] - sq)[:, None], float("-inf"))
    return torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), v)


@pytest.mark.gpu
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("rotary_frac", [0.5, 1.0])
def test_kvcache_append_rotary_vs_torch_reference(rotary_frac, interleaved, causal):
    import vllm_flash_attn as fa
    torch.manual_seed(0)
    dev, dt = "cuda", torch.float16
    b, sq, sk, h, hk, d = 3, 4, 300, 6, 2, 128
    ro = int(rotary_frac * d)
    q = torch.randn(b, sq, h, d, device=dev, dtype=dt)
    kn, vn = (torch.randn(b, sq, hk, d, device=dev, dtype=dt) for _ in range(2))
    kc, vc = (torch.randn(b, sk, hk, d, device=dev, dtype=dt) for _ in range(2))
    cache_seqlens = torch.randint(0, sk - sq + 1, (b,), dtype=torch.int32, device=dev)
    ang = torch.rand(sk, ro // 2, device=dev) * 2 * math.pi
    cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    if causal:  # query token t sits at position cache_seqlens + t
        q_ro = apply_rotary_emb(q, cos, sin, seqlen_offsets=cache_seqlens, interleaved=interleaved)
    else:       # all query tokens share position cache_seqlens (tests/test_flash_attn.py:1961-1971)
        q_ro = apply_rotary_emb(q.reshape(b, 1, sq * h, d), cos, sin, seqlen_offsets=cache_seqlens,
                                interleaved=interleaved).reshape(b, sq, h, d)
    k_ro = apply_rotary_emb(kn, cos, sin, seqlen_offsets=cache_seqlens, interleaved=interleaved)
    ar = torch.arange(sk, device=dev)[None, :]
    upd = (ar >= cache_seqlens[:, None]) & (ar < cache_seqlens[:, None] + sq)
    k_ref, v_ref = kc.clone(), vc.clone()
    k_ref[upd] = k_ro.reshape(-1, hk, d)
    v_ref[upd] = vn.reshape(-1, hk, d)
    key_mask = ar < cache_seqlens[:, None] + sq
    out_ref = _attn_ref(q_ro, k_ref, v_ref, key_mask, causal)
    out_pt = _attn_ref(q_ro, k_ref, v_ref, key_mask, causal).to(dt)  # same-dtype rounding of the reference
    out = fa.flash_attn_with_kvcache(q, kc, vc, kn, vn, rotary_cos=cos, rotary_sin=sin,
                                     cache_seqlens=cache_seqlens, causal=causal,
                                     rotary_interleaved=interleaved)
    assert torch.equal(vc, v_ref)                                             # :2074
    assert torch.allclose(kc.float(), k_ref.float(), rtol=1e-3, atol=1e-3)   # :2073
    err_pt = (out_pt.float() - out_ref).abs().max().item()
    assert (out.float() - out_ref).abs().max().item() <= 3 * max(err_pt, 2.0 ** -11) + 1e-3  # :2075-2076 (rotated q in fp16)


@pytest.mark.gpu
def test_varlen_via_unpad_pad_matches_dense_masked():
    """The reference's ragged set-up (generate_qkv, tests/test_flash_attn.py:70-175): unpad with the
    padding masks, run the varlen entry point, pad back -- against the dense call per sequence."""
    import vllm_flash_attn as fa
    torch.manual_seed(1)
    dev, dt = "cuda", torch.bfloat16
    b, s, h, d = 4, 160, 4, 64
    q, k, v = (torch.randn(b, s, h, d, device=dev, dtype=dt, requires_grad=True) for _ in range(3))
    lens = torch.tensor([160, 97, 33, 1], device=dev)
    mask = torch.arange(s, device=dev)[None, :] < lens[:, None]
    qu, iq, cq, mq = unpad_input(q, mask)
    ku, ik, ck, mk = unpad_input(k, mask)
    vu = unpad_input(v, mask)[0]
    out = pad_input(fa.flash_attn_varlen_func(qu, ku, vu, cq, ck, mq, mk, causal=True), iq, b, s)
    g = torch.randn_like(out)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
    for i, n in enumerate(lens.tolist()):
        qi, ki, vi = (x[i:i + 1, :n].detach().clone().requires_grad_() for x in (q, k, v))
        oi = fa.flash_attn_func(qi, ki, vi, causal=True)
        assert torch.equal(oi[0], out[i, :n]) and out[i, n:].abs().max().item() == 0 if n < s else True
        gi = torch.autograd.grad(oi, (qi, ki, vi), g[i:i + 1, :n])
        for a, r in zip(gi, (dq, dk, dv)):
            assert torch.equal(a[0], r[i, :n])
            assert n == s or r[i, n:].abs().max().item() == 0
