#!/usr/bin/env python3
"""A serving engine's mixed batch through flash_attn_varlen_func-style arguments: many decode
sequences (one query position each) and a few prefill chunks in ONE ragged call over a paged cache,
grouped heads (h32/8 d128 bf16, page 16, causal).  With and without the packing of grouped heads
into rows (development switch 2, bit 2) (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH
from perf_decode import med_ms

lib = cabi.lib()
h, hk, d, page = 32, 8, 128, 16
for (n_dec, kv_dec, n_pre, sq_pre, kv_pre) in ((60, 4096, 4, 512, 4096), (120, 2048, 8, 256, 2048), (250, 8192, 6, 1024, 8192),
                                                (0, 0, 32, 512, 4096), (64, 4096, 0, 0, 0)):
    torch.manual_seed(0)
    ql = [1] * n_dec + [sq_pre] * n_pre
    kl = [kv_dec] * n_dec + [kv_pre] * n_pre
    b = len(ql)
    maxblk = (max(kl) + page - 1) // page
    nblk = sum((n + page - 1) // page for n in kl)
    kp = torch.randn(nblk, page, hk, d, device="cuda", dtype=torch.bfloat16)
    vp = torch.randn_like(kp)
    bt = torch.zeros(b, maxblk, dtype=torch.int32, device="cuda")
    perm = torch.randperm(nblk, device="cuda").to(torch.int32)
    o = 0
    for i, n in enumerate(kl):
        nb = (n + page - 1) // page
        bt[i, :nb] = perm[o:o + nb]
        o += nb
    cu = lambda l: torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device="cuda")
    cq, ck = cu(ql), cu(kl)
    q = torch.randn(sum(ql), h, d, device="cuda", dtype=torch.bfloat16)
    res = []
    for flag in (4, 0):
        lib.vfa_debug_set_tuning(2, flag)
        t = med_ms(lambda: cabi.fwd(q, kp, vp, causal=True, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max(ql),
                                    max_seqlen_k=max(kl), block_table=bt))
        res.append(t * 1000)
    lib.vfa_debug_set_tuning(2, 0)
    print(f"{n_dec} x (1 | {kv_dec}) + {n_pre} x ({sq_pre} | {kv_pre}): one block per query head {res[0]:7.1f} us, packed {res[1]:7.1f} us", flush=True)
