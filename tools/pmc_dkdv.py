#!/usr/bin/env python3
"""Runs the backward of the headline shape a few times with each dK/dV kernel variant (for
rocprofv3 --pmc passes; development tool).  argv[1] = causal|full"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches
lib = cabi.lib()
causal = len(sys.argv) < 2 or sys.argv[1] == "causal"
q = torch.randn(8, 4096, 32, 128, device="cuda", dtype=torch.bfloat16)
k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
for var in (1, 2, 1, 2, 1, 2):
    lib.vfa_debug_set_tuning(0, var)
    cabi.bwd(do, q, k, v, out, lse, causal=causal)
torch.cuda.synchronize()
