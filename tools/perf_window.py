#!/usr/bin/env python3
"""Forward time under a sliding window (development tool): b2 s8192 h16 d128 bf16 causal,
window_left swept.  VFA_LIB selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi
if os.environ.get("VFA_LIB"):
    cabi.LIB_PATH = os.environ["VFA_LIB"]
torch.manual_seed(0)
b, s, h, d = 2, 8192, 16, 128
q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
k = torch.randn_like(q); v = torch.randn_like(q)
out = torch.empty_like(q)
for wl in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1536, 2048, 3072, 4096, 6144]:
    for _ in range(3):
        cabi.fwd(q, k, v, causal=True, window=(wl, 0), out=out)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            cabi.fwd(q, k, v, causal=True, window=(wl, 0), out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8)
    ts.sort()
    lse = cabi.fwd(q, k, v, causal=True, window=(wl, 0), out=out)[1]
    do = torch.randn_like(q)
    for _ in range(2):
        cabi.bwd(do, q, k, v, out, lse, causal=True, window=(wl, 0))
    tb = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            cabi.bwd(do, q, k, v, out, lse, causal=True, window=(wl, 0))
        e1.record(); torch.cuda.synchronize()
        tb.append(e0.elapsed_time(e1) / 4)
    tb.sort()
    print(f"window {wl:5d}: fwd {ts[3] * 1000:8.1f} us   bwd {tb[3] * 1000:8.1f} us")
