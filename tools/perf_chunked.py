#!/usr/bin/env python3
"""Chunked prefill (a block of new query tokens against a long KV context) on one GPU: TFLOPS of
flash_attn_func / flash_attn_with_kvcache with the split heuristic, with num_splits=1, and of torch
SDPA (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import vllm_flash_attn as fa
from seqlen_sweep import med_ms


def main():
    torch.manual_seed(0)
    h, hk, d, dt = 32, 8, 128, torch.bfloat16
    for b, sq, sk in ((1, 64, 8192), (1, 128, 8192), (1, 512, 8192), (1, 2048, 32768), (4, 256, 4096), (8, 512, 8192),
                      (16, 128, 2048), (1, 1024, 1024), (2, 4096, 4096)):
        q = torch.randn(b, sq, h, d, device="cuda", dtype=dt)
        k = torch.randn(b, sk, hk, d, device="cuda", dtype=dt)
        v = torch.randn_like(k)
        lens = torch.full((b,), sk, dtype=torch.int32, device="cuda")
        # bottom-right causal: visible pairs = sq * (sk - sq) + sq (sq + 1) / 2
        flops = 4 * b * h * d * (sq * (sk - sq) + sq * (sq + 1) / 2)
        t0 = med_ms(lambda: fa.flash_attn_func(q, k, v, causal=True), it=5)
        t1 = med_ms(lambda: fa.flash_attn_with_kvcache(q, k, v, cache_seqlens=lens, causal=True), it=5)
        t2 = med_ms(lambda: fa.flash_attn_with_kvcache(q, k, v, cache_seqlens=lens, causal=True, num_splits=1), it=5)
        print(f"b{b} sq{sq} sk{sk}: flash_attn_func {t0*1e3:7.1f} us ({flops/t0/1e9:5.0f} TF)  kvcache auto {t1*1e3:7.1f} us "
              f"({flops/t1/1e9:5.0f} TF)  kvcache num_splits=1 {t2*1e3:7.1f} us ({flops/t2/1e9:5.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
