#!/usr/bin/env python3
"""Small-batch decode through the Python API, for a kernel trace (development tool):
rocprofv3 --kernel-trace --stats -- python3 tools/decode_small.py [batch] [kv]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kv = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
h, hk, d, page = 32, 8, 128, 16
dt = torch.float16
torch.manual_seed(0)
nb = b * kv // page
kp = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
vp = torch.randn_like(kp)
bt = torch.randperm(nb, device="cuda").view(b, kv // page).to(torch.int32)
q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
for _ in range(50):
    fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True)
torch.cuda.synchronize()
