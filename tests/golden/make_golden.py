#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz.

Runs ONLY in the build container, where /root/reference is mounted: it imports the reference's own
pure-PyTorch oracle (attention_ref and helpers in /root/reference/tests/test_flash_attn.py) on CPU,
with empty stand-in modules for the absent `flash_attn` package (only names are needed at import
time; none of them is called), and records inputs + expected outputs.  Nothing from the reference
travels: the fixtures hold data only.

Each fixture stores
  q, k, v, g          16-bit inputs as uint16 bit patterns (fp32 case: float32), g = dout
  out_ref, lse_ref    fp32-upcast reference output (attention_ref(upcast=True)) and logsumexp
  dq_ref, dk_ref, dv_ref   autograd gradients of the fp32-upcast reference
  err_out_pt, err_dq_pt, err_dk_pt, err_dv_pt   max |x_pt - x_ref| of the same-dtype PyTorch
                      path (upcast=False, reorder_ops=True) -- the reference tests' tolerance unit
  meta                json: shapes, flags, window, dtype, seqlens ...
"""
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF_TEST = "/root/reference/tests/test_flash_attn.py"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


def load_reference_oracle():
    names = {
        "flash_attn": ["flash_attn_func", "flash_attn_kvpacked_func", "flash_attn_qkvpacked_func",
                       "flash_attn_varlen_func", "flash_attn_varlen_kvpacked_func",
                       "flash_attn_varlen_qkvpacked_func", "flash_attn_with_kvcache"],
        "flash_attn.bert_padding": ["pad_input", "unpad_input"],
        "flash_attn.flash_attn_interface": ["_get_block_size_n"],
        "flash_attn.layers": [],
        "flash_attn.layers.rotary": ["apply_rotary_emb"],
    }
    for mod, attrs in names.items():
        m = types.ModuleType(mod)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[mod] = m
    torch.cuda.get_device_capability = lambda *a, **k: (0, 0)
    spec = importlib.util.spec_from_file_location("ref_test_flash_attn", REF_TEST)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def to_bits(t):
    if t.dtype == torch.float32:
        return t.numpy().copy()
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def ref_lse(ref, q, k, causal, window, attn_bias, qmask=None, kmask=None):
    """logsumexp of the masked, scaled, biased scores, computed as attention_ref builds them."""
    from einops import rearrange, repeat
    if causal:
        window = (window[0], 0)
    q, k = q.float(), k.float()
    d = q.shape[-1]
    k = repeat(k, "b s h d -> b s (h g) d", g=q.shape[2] // k.shape[2])
    scores = torch.einsum("bthd,bshd->bhts", q / math.sqrt(d), k)
    if kmask is not None:
        scores.masked_fill_(rearrange(~kmask, "b s -> b 1 1 s"), float("-inf"))
    if window[0] >= 0 or window[1] >= 0:
        lm = ref.construct_local_mask(q.shape[1], k.shape[1], window, qmask, kmask, q.device)
        scores.masked_fill_(lm, float("-inf"))
    if attn_bias is not None:
        scores = scores + attn_bias
    lse = torch.logsumexp(scores, dim=-1)
    lse = torch.where(torch.isinf(lse) & (lse < 0), torch.full_like(lse, float("inf")), lse)
    return lse  # (b, h, sq)


def dense_case(ref, name, dtype, b, sq, sk, h, hk, d, causal=False, window=(-1, -1), alibi=False,
               seed=0, qlens=None, klens=None, grads=True, scale_in=1.0):
    torch.manual_seed(seed)
    tdt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    q = (torch.randn(b, sq, h, d) * scale_in).to(tdt)
    k = (torch.randn(b, sk, hk, d) * scale_in).to(tdt)
    v = (torch.randn(b, sk, hk, d) * scale_in).to(tdt)
    g = torch.randn(b, sq, h, d).to(tdt)
    qmask = kmask = None
    if qlens is not None:
        qmask = torch.arange(sq)[None, :] < torch.tensor(qlens)[:, None]
    if klens is not None:
        kmask = torch.arange(sk)[None, :] < torch.tensor(klens)[:, None]
    slopes = None
    bias = None
    if alibi:
        slopes = torch.rand(b, h, dtype=torch.float32) * 0.3
        bias = ref.attn_bias_from_alibi_slopes(slopes, sq, sk, qmask, kmask, causal=causal)
    qr, kr, vr = (x.clone().requires_grad_(grads) for x in (q, k, v))
    out_ref, _ = ref.attention_ref(qr, kr, vr, qmask, kmask, bias, 0.0, None, causal=causal,
                                   window_size=window, upcast=True)
    # keep the un-rounded fp32 reference: recompute without the final cast
    with torch.no_grad():
        out_ref32, _ = ref.attention_ref(q.float(), k.float(), v.float(), qmask, kmask, bias, 0.0,
                                         None, causal=causal, window_size=window, upcast=True)
        out_pt, _ = ref.attention_ref(q, k, v, qmask, kmask, bias, 0.0, None, causal=causal,
                                      window_size=window, upcast=False, reorder_ops=True)
    data = dict(q=to_bits(q), k=to_bits(k), v=to_bits(v), g=to_bits(g),
                out_ref=out_ref32.numpy(), lse_ref=ref_lse(ref, q, k, causal, window, bias, qmask,
                                                           kmask).numpy(),
                err_out_pt=np.float32((out_pt.float() - out_ref32).abs().max().item()))
    if slopes is not None:
        data["alibi_slopes"] = slopes.numpy()
    if grads:
        # fp32 gradients of the fp32 reference
        q32, k32, v32 = (x.float().clone().requires_grad_() for x in (q, k, v))
        o32, _ = ref.attention_ref(q32, k32, v32, qmask, kmask, bias, 0.0, None, causal=causal,
                                   window_size=window, upcast=True)
        dq, dk, dv = torch.autograd.grad(o32, (q32, k32, v32), g.float())
        qp, kp, vp = (x.clone().requires_grad_() for x in (q, k, v))
        opt, _ = ref.attention_ref(qp, kp, vp, qmask, kmask, bias, 0.0, None, causal=causal,
                                   window_size=window, upcast=False, reorder_ops=True)
        dqp, dkp, dvp = torch.autograd.grad(opt, (qp, kp, vp), g)
        data.update(dq_ref=dq.numpy(), dk_ref=dk.numpy(), dv_ref=dv.numpy(),
                    err_dq_pt=np.float32((dqp.float() - dq).abs().max().item()),
                    err_dk_pt=np.float32((dkp.float() - dk).abs().max().item()),
                    err_dv_pt=np.float32((dvp.float() - dv).abs().max().item()))
    meta = dict(name=name, kind="dense", dtype=dtype, b=b, sq=sq, sk=sk, h=h, hk=hk, d=d,
                causal=causal, window=list(window), alibi=alibi, seed=seed, qlens=qlens, klens=klens)
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **data)
    print(f"{name}: err_out_pt={data['err_out_pt']:.3e}")


def dropout_case(ref, name, dtype, b, sq, sk, h, hk, d, p_dropout, philox_seed, philox_offset,
                 causal=False, window=(-1, -1), alibi=False, seed=0):
    """Dropout: the keep mask is the reference kernel's Philox pattern (dropout.h / philox.cuh) as
    restated by oracle/fa_oracle.c for (philox_seed, philox_offset); the expected outputs and
    gradients come from the reference's attention_ref(dropout_p, dropout_mask).  This pins the
    dropout MATH given the mask; the Philox core is pinned separately (philox_aten.json)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT_DIR)))
    from oracle import fa_oracle
    torch.manual_seed(seed)
    tdt = {"fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    q = torch.randn(b, sq, h, d).to(tdt)
    k = torch.randn(b, sk, hk, d).to(tdt)
    v = torch.randn(b, sk, hk, d).to(tdt)
    g = torch.randn(b, sq, h, d).to(tdt)
    slopes = bias = None
    if alibi:
        slopes = torch.rand(b, h, dtype=torch.float32) * 0.3
        bias = ref.attn_bias_from_alibi_slopes(slopes, sq, sk, None, None, causal=causal)
    mask_np = fa_oracle.dropout_mask(philox_seed, philox_offset, b, h, sq, sk, p_dropout)
    mask = torch.from_numpy(mask_np)
    q32, k32, v32 = (x.float().clone().requires_grad_() for x in (q, k, v))
    o32, _ = ref.attention_ref(q32, k32, v32, None, None, bias, p_dropout, mask, causal=causal,
                               window_size=window, upcast=True)
    dq, dk, dv = torch.autograd.grad(o32, (q32, k32, v32), g.float())
    qp, kp, vp = (x.clone().requires_grad_() for x in (q, k, v))
    opt, _ = ref.attention_ref(qp, kp, vp, None, None, bias, p_dropout, mask, causal=causal,
                               window_size=window, upcast=False, reorder_ops=True)
    dqp, dkp, dvp = torch.autograd.grad(opt, (qp, kp, vp), g)
    data = dict(q=to_bits(q), k=to_bits(k), v=to_bits(v), g=to_bits(g),
                keep_mask=np.packbits(mask_np), out_ref=o32.detach().numpy(),
                lse_ref=ref_lse(ref, q, k, causal, window, bias).numpy(),
                err_out_pt=np.float32((opt.float() - o32).abs().max().item()),
                dq_ref=dq.numpy(), dk_ref=dk.numpy(), dv_ref=dv.numpy(),
                err_dq_pt=np.float32((dqp.float() - dq).abs().max().item()),
                err_dk_pt=np.float32((dkp.float() - dk).abs().max().item()),
                err_dv_pt=np.float32((dvp.float() - dv).abs().max().item()))
    if slopes is not None:
        data["alibi_slopes"] = slopes.numpy()
    meta = dict(name=name, kind="dropout", dtype=dtype, b=b, sq=sq, sk=sk, h=h, hk=hk, d=d,
                causal=causal, window=list(window), alibi=alibi, seed=seed, p_dropout=p_dropout,
                philox_seed=philox_seed, philox_offset=philox_offset)
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **data)
    print(f"{name}: err_out_pt={data['err_out_pt']:.3e} keep={mask_np.mean():.4f}")


def kvcache_case(ref, name, dtype, b, sq, sk, h, hk, d, causal, page, seed=0, alibi=False,
                 window=(-1, -1)):
    """Read-only kv-cache decode: cache_seqlens < capacity; optional paged cache."""
    torch.manual_seed(seed)
    tdt = {"fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    q = torch.randn(b, sq, h, d).to(tdt)
    data = {}
    if page is None:
        k_cache = torch.randn(b, sk, hk, d).to(tdt)
        v_cache = torch.randn(b, sk, hk, d).to(tdt)
    else:
        k_cache, v_cache, block_table, k_paged, v_paged, _ = ref._generate_block_kvcache(
            sk, page, b, hk, d, "cpu", tdt)
        data.update(block_table=block_table.numpy(), k_paged=to_bits(k_paged),
                    v_paged=to_bits(v_paged))
    cache_seqlens = torch.randint(1, sk + 1, (b,), dtype=torch.int32)
    kmask = torch.arange(sk)[None, :] < cache_seqlens[:, None]
    slopes = bias = None
    if alibi:
        slopes = torch.rand(b, h, dtype=torch.float32) * 0.3
        bias = ref.attn_bias_from_alibi_slopes(slopes, sq, sk, None, kmask, causal=causal)
        data["alibi_slopes"] = slopes.numpy()
    with torch.no_grad():
        out_ref, _ = ref.attention_ref(q.float(), k_cache.float(), v_cache.float(), None, kmask,
                                       bias, 0.0, None, causal=causal, window_size=window)
        out_pt, _ = ref.attention_ref(q, k_cache, v_cache, None, kmask, bias, 0.0, None,
                                      causal=causal, window_size=window, upcast=False,
                                      reorder_ops=True)
    data.update(q=to_bits(q), k=to_bits(k_cache), v=to_bits(v_cache),
                cache_seqlens=cache_seqlens.numpy(), out_ref=out_ref.numpy(),
                lse_ref=ref_lse(ref, q, k_cache, causal, window, bias, None, kmask).numpy(),
                err_out_pt=np.float32((out_pt.float() - out_ref).abs().max().item()))
    meta = dict(name=name, kind="kvcache", dtype=dtype, b=b, sq=sq, sk=sk, h=h, hk=hk, d=d,
                causal=causal, window=list(window), alibi=alibi, page=page, seed=seed)
    data["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **data)
    print(f"{name}: err_out_pt={data['err_out_pt']:.3e}")


def dropout_cases(ref):
    # dropout_p = 0.17 is the reference tests' value (tests/test_flash_attn.py:549,662,850)
    dropout_case(ref, "drop_bf16_d128_s128_causal", "bf16", 2, 128, 128, 2, 2, 128, 0.17, 0, 0, causal=True)
    dropout_case(ref, "drop_fp16_d64_113x203_full", "fp16", 1, 113, 203, 2, 2, 64, 0.17, 1234567, 96)
    dropout_case(ref, "drop_bf16_d128_gqa_local", "bf16", 1, 200, 263, 4, 2, 128, 0.3, 42, 1 << 33,
                 window=(57, 21))
    dropout_case(ref, "drop_fp16_d32_alibi_causal", "fp16", 2, 96, 160, 2, 1, 32, 0.5, 7, 64,
                 causal=True, alibi=True)


def main():
    ref = load_reference_oracle()
    if len(sys.argv) > 1 and sys.argv[1] == "--dropout-only":
        return dropout_cases(ref)
    dropout_cases(ref)
    # (1) plumbing configuration: fp32, also checked against torch SDPA in the test
    dense_case(ref, "c1_fp32_b2_s128_h4_d64", "fp32", 2, 128, 128, 4, 4, 64, grads=False)
    # (2) d x dtype x causal at the reference tests' odd shapes (tests/test_flash_attn.py:862-876)
    dense_case(ref, "bf16_d128_s128_causal", "bf16", 1, 128, 128, 2, 2, 128, causal=True)
    dense_case(ref, "bf16_d128_s128_full", "bf16", 1, 128, 128, 1, 1, 128)
    dense_case(ref, "fp16_d128_113x203_causal", "fp16", 1, 113, 203, 1, 1, 128, causal=True)
    dense_case(ref, "fp16_d64_113x203_full", "fp16", 2, 113, 203, 1, 1, 64)
    dense_case(ref, "bf16_d64_256x512_causal", "bf16", 1, 256, 512, 1, 1, 64, causal=True)
    dense_case(ref, "bf16_d128_512x256_causal", "bf16", 1, 512, 256, 1, 1, 128, causal=True, grads=False)  # sq>sk: fully masked rows
    dense_case(ref, "fp16_d128_1x239_full", "fp16", 2, 1, 239, 2, 2, 128, grads=False)
    dense_case(ref, "bf16_d64_s384_full", "bf16", 1, 384, 384, 1, 1, 64)
    # (3) GQA / MQA
    dense_case(ref, "bf16_d128_gqa4x2_causal", "bf16", 1, 160, 160, 4, 2, 128, causal=True)
    dense_case(ref, "fp16_d64_mqa_full", "fp16", 1, 97, 131, 3, 1, 64)
    # (4) local windows, two-sided and one-sided
    dense_case(ref, "bf16_d128_local_17_5", "bf16", 1, 200, 263, 1, 1, 128, window=(17, 5))
    dense_case(ref, "bf16_d128_local_64_0", "bf16", 1, 256, 256, 2, 1, 128, window=(64, 0), grads=False)
    dense_case(ref, "fp16_d64_local_m1_30", "fp16", 1, 128, 217, 1, 1, 64, window=(-1, 30))
    # (5) ALiBi causal / non-causal / with local
    dense_case(ref, "bf16_d128_alibi_causal", "bf16", 2, 96, 160, 2, 2, 128, causal=True, alibi=True)
    dense_case(ref, "fp16_d64_alibi_full", "fp16", 2, 113, 203, 2, 2, 64, alibi=True, grads=False)
    dense_case(ref, "bf16_d128_alibi_local_gqa", "bf16", 2, 192, 192, 2, 1, 128, causal=True,
               window=(48, 0), alibi=True, grads=False)
    # (6) ragged batches (padding masks -> cu_seqlens in the tests); literal lengths of
    #     tests/test_flash_attn.py:2299 ([0,76,110,256] -> 76, 34, 146)
    dense_case(ref, "bf16_d64_varlen_76_34_146_causal", "bf16", 3, 146, 146, 1, 1, 64, causal=True,
               qlens=[76, 34, 146], klens=[76, 34, 146])
    dense_case(ref, "fp16_d128_varlen_ragged_gqa", "fp16", 3, 128, 200, 2, 1, 128,
               qlens=[128, 57, 1], klens=[200, 111, 64], grads=False)
    dense_case(ref, "bf16_d128_varlen_local_alibi", "bf16", 2, 160, 160, 2, 1, 128, causal=True,
               window=(32, 0), alibi=True, qlens=[160, 93], klens=[160, 93], grads=False)
    # other head dims of the fork (multiples of 32)
    dense_case(ref, "bf16_d32_s96_causal", "bf16", 1, 96, 96, 2, 2, 32, causal=True)
    dense_case(ref, "fp16_d96_s80x144_full", "fp16", 1, 80, 144, 1, 1, 96, grads=False)
    dense_case(ref, "bf16_d256_s64x96_causal", "bf16", 1, 64, 96, 1, 1, 256, causal=True, grads=False)
    # overflow regression inputs scaled x5 (tests/test_flash_attn.py:2183-2228)
    dense_case(ref, "fp16_d64_s17_x5_causal", "fp16", 2, 17, 17, 2, 2, 64, causal=True, scale_in=5.0)
    dense_case(ref, "fp16_d64_s5_x5_causal", "fp16", 2, 5, 5, 2, 2, 64, causal=True, scale_in=5.0)
    dense_case(ref, "bf16_d128_s128_x3_causal", "bf16", 1, 128, 128, 1, 1, 128, causal=True, scale_in=3.0)
    # (7) kv-cache decode: contiguous and paged
    kvcache_case(ref, "kvc_fp16_d128_gqa_sq1", "fp16", 3, 1, 300, 4, 1, 128, True, None)
    kvcache_case(ref, "kvc_bf16_d64_sq3_causal", "bf16", 2, 3, 257, 2, 2, 64, True, None)
    kvcache_case(ref, "kvc_fp16_d128_paged16", "fp16", 2, 1, 200, 4, 1, 128, False, 16)
    kvcache_case(ref, "kvc_bf16_d64_paged256_sq2", "bf16", 2, 2, 300, 2, 1, 64, True, 256)
    kvcache_case(ref, "kvc_fp16_d128_alibi_sq1", "fp16", 2, 1, 180, 2, 1, 128, True, None, alibi=True)

if __name__ == "__main__":
    main()
