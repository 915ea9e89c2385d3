#!/usr/bin/env python3
"""Decode K+V bandwidth against the cache layout and the split count (development tool): the same
bytes read as 256-byte rows at a 2 KiB stride (b, kv, hk, d), as one contiguous stream per head
(b, hk, kv, d) viewed through strides, and with the key range cut over 1..4 workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa
from perf_decode import med_ms


def main():
    torch.manual_seed(0)
    b, kv, h, hk, d = 128, 8192, 32, 8, 128
    dt = torch.float16
    q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
    lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
    gb = 2.0 * b * kv * hk * d * 2 / 1e9
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=dt)
    vc = torch.randn_like(kc)
    out = torch.empty_like(q)
    for rep in range(2):
        for ns in (0, 1, 2, 4):
            t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, out=out, num_splits=ns))
            print(f"(b,kv,hk,d) num_splits={ns}: {t:.3f} ms  {gb / t * 1000:.0f} GB/s", flush=True)
    fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, out=out, num_splits=1)
    ref = out.clone()
    kh = kc.transpose(1, 2).contiguous().transpose(1, 2)   # (b, kv, hk, d) view of (b, hk, kv, d) storage
    vh = vc.transpose(1, 2).contiguous().transpose(1, 2)
    for ns in (1, 2):
        t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kh, vh, cache_seqlens=lens, causal=True, out=out, num_splits=ns))
        print(f"(b,hk,kv,d) storage num_splits={ns}: {t:.3f} ms  {gb / t * 1000:.0f} GB/s  diff {(out.float()-ref.float()).abs().max().item():.1e}", flush=True)
    # plain read of the same bytes by a library kernel, for scale
    t = med_ms(lambda: (torch.ops.aten.sum(kc.view(torch.int16).view(-1)), torch.ops.aten.sum(vc.view(torch.int16).view(-1))))
    print(f"aten.sum over K and V: {t:.3f} ms  {gb / t * 1000:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
