#!/usr/bin/env python3
"""Development tool: where the two-phase dK/dV kernel (development switch 0 = 3) differs from the
8-wave kernel (= 4): max |diff| per head and per 32-key group."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches
lib = cabi.lib()
b, sq, sk, h, hk, d = [int(x) for x in (sys.argv[1:7] if len(sys.argv) > 6 else (1, 256, 512, 2, 2, 128))]
causal = len(sys.argv) > 7 and sys.argv[7] == "causal"
torch.manual_seed(0)
q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.bfloat16)
k = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
v = torch.randn(b, sk, hk, d, device="cuda", dtype=torch.bfloat16)
do = torch.randn_like(q)
out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
res = {}
for var in (4, 3):
    lib.vfa_debug_set_tuning(0, var)
    res[var] = [t.float() for t in cabi.bwd(do, q, k, v, out, lse, causal=causal)[:3]]
    torch.cuda.synchronize()
lib.vfa_debug_set_tuning(0, 0)
for name, i in (("dk", 1), ("dv", 2)):
    diff = (res[4][i] - res[3][i]).abs()  # (b, sk, hk, d)
    print(name, "max", diff.max().item(), "ref max", res[4][i].abs().max().item())
    per = diff.amax(dim=(0, 3))  # (sk, hk)
    g = per.view(-1, 32, hk).amax(dim=1) if sk % 32 == 0 else per
    for hh in range(hk):
        print("  head", hh, " ".join(f"{x:.2g}" for x in g[:, hh].tolist()))
