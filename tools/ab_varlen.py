#!/usr/bin/env python3
"""Interleaved timing, in one process, of ragged forward calls from several builds of the library
(development tool):  ab_varlen.py lib1.so lib2.so ...   Cases: the ragged benchmark batch (GQA,
causal, window 1024, ALiBi), the same without ALiBi / without window, and a paged prefill."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cabi
from perf_decode import med_ms

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
handles = [cabi.load(os.path.abspath(p)) for p in libs]
cu = lambda l: torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device="cuda")
SEQ = [8192, 6144, 4096, 3072, 2048, 1024, 513, 127]
CASES = [("c4: window 1024 + ALiBi", 128, dict(causal=True, window=(1024, 0)), True),
         ("causal + ALiBi", 128, dict(causal=True), True),
         ("causal, window 1024", 128, dict(causal=True, window=(1024, 0)), False),
         ("causal", 128, dict(causal=True), False),
         ("causal d64", 64, dict(causal=True), False),
         ("causal + ALiBi d64", 64, dict(causal=True), True)]
for name, d, kw, alibi in CASES:
    torch.manual_seed(0)
    h, hk = 32, 8
    q = torch.randn(sum(SEQ), h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(SEQ), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    c = cu(SEQ)
    sl = torch.rand(len(SEQ), h, device="cuda") * 0.3 if alibi else None
    outs, ts = [], [[] for _ in handles]
    for rnd in range(3):
        for i, hd in enumerate(handles):
            cabi._lib = hd
            f = lambda: cabi.fwd(q, k, v, cu_seqlens_q=c, cu_seqlens_k=c, max_seqlen_q=max(SEQ), max_seqlen_k=max(SEQ),
                                 alibi_slopes=sl, **kw)
            if rnd == 0:
                outs.append(f()[0])
            ts[i].append(med_ms(f))
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name}: " + "   ".join(f"{os.path.basename(p)} {min(t):.3f} ms" for p, t in zip(libs, ts)) + ("" if same else "   OUTPUTS DIFFER"),
          flush=True)
