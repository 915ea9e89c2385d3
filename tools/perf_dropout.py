#!/usr/bin/env python3
"""Forward and backward time with dropout (b8 s4096 h32 d128 bf16 causal, p = 0.1).  Development tool."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, cabi
torch.manual_seed(0)
b, s, h, d = 8, 4096, 32, 128
q, k, v, do = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(4))
rng = torch.zeros(2, dtype=torch.int64, device="cuda")
def fwd():
    return cabi.fwd(q, k, v, causal=True, p_dropout=0.1, philox_seed=1, philox_offset=0, rng_state=rng)
out, lse = fwd()
def bwd():
    return cabi.bwd(do, q, k, v, out, lse, causal=True, p_dropout=0.1, rng_state=rng)
for name, fn, fl in (("fwd", fwd, 1.0), ("bwd", bwd, 2.5)):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(name, round(t, 3), "ms", round(fl * 4 * b * h * s * s * d / 2 / t / 1e9), "TF")
