#!/usr/bin/env python3
"""Forward and backward TFLOP/s per head dim (development tool): b4 s4096 h16 bf16, causal and not,
this library through the C ABI; times are medians of HIP-event brackets around 5 queued calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, cabi


def med_ms(fn, n=7, inner=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


dims = [int(a) for a in sys.argv[1:]] or [64, 96, 128, 160, 192, 256]
for d in dims:
    for causal in (True, False):
        b, s, h = 4, 4096, 16
        q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
        k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
        out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
        f = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
        tf = med_ms(lambda: cabi.fwd(q, k, v, causal=causal))
        tb = med_ms(lambda: cabi.bwd(do, q, k, v, out, lse, causal=causal))
        print(f"d{d} causal={causal}: fwd {tf:.3f} ms {f / tf / 1e9:.0f} TF   bwd {tb:.3f} ms {2.5 * f / tb / 1e9:.0f} TF", flush=True)
