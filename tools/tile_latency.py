#!/usr/bin/env python3
"""Forward time of small launches (one to three workgroups per CU), per K/V tile of the longest
block: what a block costs when nothing else hides its dependent chain (development tool).
Optional argument: a development switch K:V to set (e.g. 4:1 = the 64-rows-per-wave forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch, cabi
from perf_decode import med_ms
tune = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:] if ":" in a]
for (b, s, h, d, causal) in ((1, 4096, 8, 64, False), (1, 4096, 16, 64, False), (1, 4096, 24, 64, False), (1, 4096, 8, 128, False),
                             (1, 4096, 16, 128, False), (1, 4096, 32, 128, False), (1, 4096, 32, 128, True), (4, 2048, 16, 64, True)):
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    res = []
    for sw in [None] + tune:
        if sw is None:
            t = med_ms(lambda: cabi.fwd(q, k, v, causal=causal))
        else:
            with cabi.dev(*sw):
                t = med_ms(lambda: cabi.fwd(q, k, v, causal=causal))
        res.append(t)
    tiles = s // 64
    print(f"b{b} s{s} h{h} d{d} causal={causal}: " + "  ".join(f"{t*1e3:.1f} us ({t*1e3/tiles:.2f} per tile)" for t in res), flush=True)
