#!/usr/bin/env python3
"""TFLOPS vs sequence length on one GPU, the sweep of the reference's own benchmark
(benchmarks/benchmark_flash_attention.py:65-76: 16k tokens per batch, model dim 2048, head dim 64 /
128, causal and not, forward / backward / both), next to torch SDPA on the same inputs.
Writes one JSON document (default profiles/r01_seqlen_sweep.json)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import cabi


def med_ms(fn, warm=3, it=7, reps=8):
    """Median over `it` measurements of the mean time of `reps` back-to-back calls: with several
    calls queued the GPU does not wait for the host between them, so the host-side cost of a call
    (argument set-up, output allocation) is not billed to kernels that run for tens of microseconds."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()  # (a call in flight while the timed ones are queued)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r01_seqlen_sweep.json")
    torch.manual_seed(0)
    rows = []
    for d in (64, 128):
        h = 2048 // d
        for causal in (False, True):
            for s in (512, 1024, 2048, 4096, 8192, 16384):
                b = 16384 // s
                q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
                k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
                out, lse = cabi.fwd(q, k, v, causal=causal)[:2]
                flops = 4 * b * h * s * s * d / (2 if causal else 1)
                t_f = med_ms(lambda: cabi.fwd(q, k, v, causal=causal, out=out))
                t_b = med_ms(lambda: cabi.bwd(do, q, k, v, out, lse, causal=causal))
                qt, kt, vt = (x.transpose(1, 2).detach().requires_grad_() for x in (q, k, v))
                t_sf = med_ms(lambda: F.scaled_dot_product_attention(qt, kt, vt, is_causal=causal))
                o2 = F.scaled_dot_product_attention(qt, kt, vt, is_causal=causal)
                g = do.transpose(1, 2)
                t_sb = med_ms(lambda: torch.autograd.grad(o2, (qt, kt, vt), g, retain_graph=True))
                r = {"headdim": d, "nheads": h, "batch": b, "seqlen": s, "causal": causal,
                     "fwd_tflops": flops / t_f / 1e9, "bwd_tflops": 2.5 * flops / t_b / 1e9,
                     "fwd_bwd_tflops": 3.5 * flops / (t_f + t_b) / 1e9,
                     "sdpa_fwd_tflops": flops / t_sf / 1e9, "sdpa_bwd_tflops": 2.5 * flops / t_sb / 1e9,
                     "sdpa_fwd_bwd_tflops": 3.5 * flops / (t_sf + t_sb) / 1e9}
                rows.append(r)
                print(json.dumps({k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in r.items()}), flush=True)
    with open(out_path, "w") as f:
        json.dump({"note": "bf16, 16k tokens per batch, dim 2048; this library through the C ABI vs torch SDPA on the "
                           "same tensors; each figure is the median of 7 timings of 8 queued calls; MI355X",
                   "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
