#!/usr/bin/env python3
"""Interleaved timing, in one process, of the decode forward (b128 sq1 kv8192 h32/hk8 d128 fp16,
query heads folded into rows as the extension does) from several builds of the library
(development tool):  ab_decode.py lib1.so lib2.so ... [--splits=1,4] [--tune=K:V ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cabi


def main():
    args = sys.argv[1:]
    tune = [tuple(int(x) for x in a.split("=", 1)[1].split(":")) for a in args if a.startswith("--tune")]
    splits = [int(x) for a in args if a.startswith("--splits") for x in a.split("=", 1)[1].split(",")] or [1]
    shape = [int(x) for a in args if a.startswith("--shape") for x in a.split("=", 1)[1].split(",")] or [128, 8192]
    libs = [a for a in args if a.endswith(".so")]
    handles = [cabi.load(os.path.abspath(p)) for p in libs]
    for h in handles:
        for k, v in tune:
            h.vfa_debug_set_tuning(k, v)
    torch.manual_seed(0)
    b, kv = shape
    g, hk, d = 4, 8, 128
    q = torch.randn(b, g, hk, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
    gb = 2.0 * b * kv * hk * d * 2 / 1e9
    for ns in splits:
        times = [[] for _ in handles]
        for rnd in range(12):
            for i, h in enumerate(handles):
                cabi._lib = h
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                cabi.fwd(q, kc, vc, cache_seqlens=lens, kvcache=True, num_splits=ns)
                e1.record()
                torch.cuda.synchronize()
                if rnd >= 3:
                    times[i].append(e0.elapsed_time(e1))
        for i, p in enumerate(libs):
            t = sorted(times[i])[len(times[i]) // 2]
            print(f"splits {ns:2d} {os.path.basename(p):28s} {t:.3f} ms  {gb / t * 1000:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
