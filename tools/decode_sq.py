#!/usr/bin/env python3
"""K+V rate of flash_attn_with_kvcache against the number of query tokens per sequence (speculative
decoding / small chunks; b64 kv4096 h32/hk8 d128 fp16 causal) (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa
from perf_decode import med_ms

b, kv, h, hk, d = 64, 4096, 32, 8, 128
torch.manual_seed(0)
kc = torch.randn(b, kv, hk, d, device="cuda", dtype=torch.float16)
vc = torch.randn_like(kc)
lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
gb = 2.0 * b * kv * hk * d * 2 / 1e9
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches
flags = [int(a) for a in sys.argv[1:]] or [0]
for sq in (1, 2, 4, 8, 16, 32, 64, 128):
    q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.float16)
    for f in flags:
        cabi.lib().vfa_debug_set_tuning(2, f)
        t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True))
        print(f"sq={sq} flags={f}: {t * 1000:.0f} us  {gb / t * 1000:.0f} GB/s of K+V", flush=True)
cabi.lib().vfa_debug_set_tuning(2, 0)
