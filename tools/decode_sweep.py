#!/usr/bin/env python3
"""Decode K+V rate of the shipped library against the split count over several (batch, cache
length) shapes, h32/hk8 d128 fp16 (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi
import bench

hip = bench.Hip()
lib = cabi.lib()
stream = torch.cuda.current_stream().cuda_stream
g, hk, d = 4, 8, 128
for b, kv in ((128, 8192), (256, 4096), (64, 16384), (32, 32768), (128, 2048), (512, 1024), (16, 8192), (128, 8000)):
    torch.manual_seed(0)
    q = torch.randn(b, g, hk, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
    gb = 2.0 * b * kv * hk * d * 2 / 1e9
    row = []
    for ns in (1, 2, 3, 4, 6, 8):
        ts = []
        for it in range(14):
            e0, e1 = hip.event(), hip.event()
            # the launch is enqueued behind a short sleep kernel so that the host-side set-up of the
            # call does not show up between the two events
            torch.cuda._sleep(200000)
            hip.record(e0, stream)
            cabi.fwd(q, kc, vc, cache_seqlens=lens, kvcache=True, num_splits=ns)
            hip.record(e1, stream)
            torch.cuda.synchronize()
            if it >= 4:
                ts.append(hip.elapsed_ms(e0, e1))
        t = sorted(ts)[len(ts) // 2]
        row.append(f"s{ns}: {t * 1000:.0f} us {gb / t * 1000:.0f}")
    print(f"b{b} kv{kv}: " + " | ".join(row), flush=True)
    del q, kc, vc
