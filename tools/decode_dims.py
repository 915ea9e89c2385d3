#!/usr/bin/env python3
"""Decode K+V rate by head dim (b64 kv4096 h32/hk8 fp16) through the Python API (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa
from perf_decode import med_ms

shapes = [(64, 4096, d) for d in (64, 96, 128, 160, 192, 256)] + [(128, 8192, 64), (128, 8192, 128), (128, 8192, 256), (64, 8192, 256)]
for b, kv, d in shapes:
    torch.manual_seed(0)
    h, hk = 32, 8
    q = torch.randn(b, 1, h, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
    gb = 2.0 * b * kv * hk * d * 2 / 1e9
    t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True))
    print(f"b{b} kv{kv} d={d}: {t * 1000:.0f} us  {gb / t * 1000:.0f} GB/s", flush=True)
