#!/usr/bin/env python3
"""Decode (seqlen_q = 1) over batch x cache length, paged cache (page 16) through the Python
drop-in API with its own split planning: K+V bandwidth and time (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa
from perf_decode import med_ms

h, hk, d, page = 32, 8, 128, 16
dt = torch.float16
torch.manual_seed(0)
for b in (1, 4, 16, 64, 256):
    for kv in (1024, 8192, 65536):
        if b * kv > 256 * 8192 * 2:
            continue
        nb = b * kv // page
        kp = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
        vp = torch.randn_like(kp)
        bt = torch.randperm(nb, device="cuda").view(b, kv // page).to(torch.int32)
        q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
        lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
        out = torch.empty_like(q)
        t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True, out=out))
        gb = 2.0 * b * kv * hk * d * 2 / 1e9
        print(f"b {b:4d} kv {kv:6d}: {t * 1000:8.1f} us  {gb / t * 1000:7.0f} GB/s", flush=True)
        del kp, vp
