#!/usr/bin/env python3
"""Decode (seqlen_q = 1) K+V bandwidth on one GPU: contiguous cache vs paged cache at several
page sizes, through the Python drop-in API (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vllm_flash_attn as fa


def med_ms(fn, warm=5, it=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    torch.manual_seed(0)
    b, kv, h, hk, d = 128, 8192, 32, 8, 128
    dt = torch.float16
    q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
    lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
    gb = 2.0 * b * kv * hk * d * 2 / 1e9
    kc = torch.randn(b, kv, hk, d, device="cuda", dtype=dt)
    vc = torch.randn_like(kc)
    out = torch.empty_like(q)
    t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, out=out))
    ref = out.clone()
    print(f"contiguous: {t:.3f} ms  {gb / t * 1000:.0f} GB/s", flush=True)
    for page in (16, 32, 64, 256):
        nb = b * kv // page
        perm = torch.randperm(nb, device="cuda")
        bt = perm.view(b, kv // page).to(torch.int32)
        kp = torch.empty(nb, page, hk, d, device="cuda", dtype=dt)
        vp = torch.empty_like(kp)
        kp[perm] = kc.view(nb, page, hk, d)
        vp[perm] = vc.view(nb, page, hk, d)
        t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kp, vp, cache_seqlens=lens, block_table=bt, causal=True, out=out))
        err = (out.float() - ref.float()).abs().max().item()
        print(f"paged, page {page}: {t:.3f} ms  {gb / t * 1000:.0f} GB/s  (max diff vs contiguous {err:.1e})", flush=True)
        del kp, vp


def prefill_paged(d=128):
    """Causal prefill over a paged cache (flash_attn_varlen_func + block_table) vs contiguous K/V."""
    torch.manual_seed(0)
    b, s, h, hk = (8, 4096, 32, 8) if d <= 128 else (4, 4096, 16, 4)
    dt = torch.bfloat16
    q = torch.randn(b * s, h, d, device="cuda", dtype=dt)
    k = torch.randn(b * s, hk, d, device="cuda", dtype=dt)
    v = torch.randn_like(k)
    cu = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device="cuda")
    flops = 4 * b * h * s * s * d / 2
    t = med_ms(lambda: fa.flash_attn_varlen_func(q, k, v, cu, cu, s, s, causal=True))
    ref = fa.flash_attn_varlen_func(q, k, v, cu, cu, s, s, causal=True)
    print(f"prefill contiguous: {t:.3f} ms  {flops / t / 1e9:.0f} TF", flush=True)
    import cabi
    cabi.LIB_PATH = cabi.DEV_LIB_PATH   # the development build: kernel-variant switches
    for page in (16, 256):
        nb = b * s // page
        perm = torch.randperm(nb, device="cuda")
        bt = perm.view(b, s // page).to(torch.int32)
        kp = torch.empty(nb, page, hk, d, device="cuda", dtype=dt)
        vp = torch.empty_like(kp)
        kp[perm] = k.view(nb, page, hk, d)
        vp[perm] = v.view(nb, page, hk, d)
        f = lambda: fa.flash_attn_varlen_func(q, kp, vp, cu, cu, s, s, causal=True, block_table=bt)
        for sw in (0, 1):   # development switch 5: 1 = paged tiles straight into LDS at every head dim
            cabi.lib().vfa_debug_set_tuning(5, sw)
            t = med_ms(f)
            err = (f().float() - ref.float()).abs().max().item()
            print(f"prefill paged, page {page}, switch {sw}: {t:.3f} ms  {flops / t / 1e9:.0f} TF  (max diff {err:.1e})", flush=True)
        cabi.lib().vfa_debug_set_tuning(5, 0)


def small_batch():
    """Decode at small batch / long context: the split-KV path and its combine kernel."""
    torch.manual_seed(0)
    h, hk, d, dt = 32, 8, 128, torch.float16
    for b, kv in ((1, 8192), (1, 32768), (1, 131072), (4, 32768), (16, 8192), (32, 4096), (64, 2048)):
        q = torch.randn(b, 1, h, d, device="cuda", dtype=dt)
        kc = torch.randn(b, kv, hk, d, device="cuda", dtype=dt)
        vc = torch.randn_like(kc)
        lens = torch.full((b,), kv, dtype=torch.int32, device="cuda")
        gb = 2.0 * b * kv * hk * d * 2 / 1e9
        t = med_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True))
        t1 = med_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, causal=True, num_splits=1))
        print(f"b{b} kv{kv}: heuristic splits {t * 1000:.1f} us ({gb / t * 1000:.0f} GB/s)   num_splits=1 {t1 * 1000:.1f} us "
              f"({gb / t1 * 1000:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "prefill":
        prefill_paged(int(sys.argv[2]) if len(sys.argv) > 2 else 128)
    elif len(sys.argv) > 1 and sys.argv[1] == "small":
        small_batch()
    else:
        main()
