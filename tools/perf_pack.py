#!/usr/bin/env python3
"""A few query positions per sequence with grouped heads against a long KV cache (chunked prefill,
draft-token verification): the prefill kernel with the group packed into the rows of its KV head
against what the shapes took before (32-row decode blocks up to 32 positions, one 128-row block
per query head beyond; development switch 2, bit 2), through the C ABI with the planner's split
count for each (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH
from perf_decode import med_ms

lib = cabi.lib()
cus = torch.cuda.get_device_properties(0).multi_processor_count
d, page = 128, 16
print("batch kv heads sq | before us (splits) | packed us (splits)")
for (b, kv) in ((1, 8192), (4, 8192), (16, 8192), (64, 4096)):
    for (h, hk) in ((32, 8), (64, 8)):
        torch.manual_seed(0)
        nb = b * kv // page
        kp = torch.randn(nb, page, hk, d, device="cuda", dtype=torch.float16)
        vp = torch.randn_like(kp)
        bt = torch.randperm(nb, device="cuda").view(b, kv // page).to(torch.int32)
        lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
        for sq in (4, 8, 16, 32, 64):
            q = torch.randn(b, sq, h, d, device="cuda", dtype=torch.float16)
            res = []
            for flag in (4, 0):
                lib.vfa_debug_set_tuning(2, flag)
                ns = lib.vfa_decode_num_splits(b, h, hk, d, kv, sq, cus)
                t = med_ms(lambda: cabi.fwd(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True, kvcache=True,
                                            num_splits=ns))
                res.append((t * 1000, ns))
            lib.vfa_debug_set_tuning(2, 0)
            print(f"b{b} kv{kv} h{h}/{hk} sq{sq}: {res[0][0]:7.1f} ({res[0][1]:2d})  {res[1][0]:7.1f} ({res[1][1]:2d})", flush=True)
