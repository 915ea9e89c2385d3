#!/usr/bin/env python3
"""Forward + backward at short sequence lengths, for a kernel trace (development tool):
rocprofv3 --kernel-trace --stats -- python3 tools/short_seq.py [seqlen] [d] [causal]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi

s = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
causal = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
b, h = 16384 // s, 2048 // d
torch.manual_seed(0)
q, k, v, do = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(4))
out, lse = cabi.fwd(q, k, v, causal=causal)
for _ in range(30):
    out, lse = cabi.fwd(q, k, v, causal=causal)
    cabi.bwd(do, q, k, v, out, lse, causal=causal)
torch.cuda.synchronize()
flops = 4.0 * b * h * s * s * d * (0.5 if causal else 1.0)
print(f"b{b} s{s} h{h} d{d} causal={causal}: fwd {flops / 1e9:.1f} GFLOP, bwd {2.5 * flops / 1e9:.1f} GFLOP")
