#!/usr/bin/env python3
"""Interleaved timing of small-batch paged decode through the Python API under two settings of a
development switch (development tool): ab_decode_small.py SWITCH VALUE_A VALUE_B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa
import cabi
cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches

sw, va, vb = (int(x) for x in sys.argv[1:4])
h, hk, d, page = 32, 8, 128, 16
dt = torch.float16
torch.manual_seed(0)
for b, kv in ((1, 8192), (1, 65536), (4, 8192), (16, 1024), (64, 1024)):
    nb = b * kv // page
    kp = torch.randn(nb, page, hk, d, device="cuda", dtype=dt)
    vp = torch.randn_like(kp)
    bt = torch.randperm(nb, device="cuda").view(b, kv // page).to(torch.int32)
    q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
    lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
    f = lambda: fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True)
    g = torch.cuda.CUDAGraph()   # replayed graphs: the GPU time of the two kernels, no host gaps
    times = {va: [], vb: []}
    graphs = {}
    for v in (va, vb):
        cabi.lib().vfa_debug_set_tuning(sw, v)
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                f()
        graphs[v] = gr
    for rnd in range(9):
        for v in (va, vb):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graphs[v].replay(); e1.record(); torch.cuda.synchronize()
            if rnd >= 2:
                times[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    print(f"b{b} kv{kv}: switch {sw} = {va}: {med(times[va]):.1f} us   = {vb}: {med(times[vb]):.1f} us", flush=True)
cabi.lib().vfa_debug_set_tuning(sw, 0)
