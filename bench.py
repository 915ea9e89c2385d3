#!/usr/bin/env python3
"""Headline benchmark: attention forward+backward TFLOPS at the headline configuration
(batch 8, seqlen 4096, 32 heads, headdim 128, bf16, causal) on N MI355X GPUs.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one forward + one backward pass of the attention op over one synthetic batch
(torch.randn inputs resident in HBM), called through the C ABI of libvfa_hip.so.  One process per
GPU: under torch.distributed.run the ranks come from the environment; started plainly with
--gpus N > 1 this script starts N ranks of itself (before it touches a GPU) and relays rank 0's
line.  --scaling weak (default): every rank runs the full per-GPU batch; --scaling strong: the
global batch is sharded contiguously over the ranks (shard_range; ragged batches are cut at
sequence boundaries balancing the visible query-key pairs, balanced_varlen_split).  No collective
touches the data path (batch entries are independent); the process group carries the timing
barrier, the max of the elapsed time and the sum of the work.  Rank 0 prints ONE JSON line.  FLOP
convention: the reference's benchmarks/benchmark_flash_attention.py:27-30 (4*b*s^2*h*d, /2 causal;
backward = 2.5x forward).

Other workloads (--workload): c2 (b4 s2048 h16 d64 causal fwd), c3_fwd, c5 (kv-cache decode, GB/s).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_TFLOPS_BF16 = 2500.0   # MI355X dense bf16/fp16 MFMA peak (spec)
PEAK_HBM_GBS = 8000.0       # MI355X HBM3E peak (spec)

WORKLOADS = {
    # name: (b, sq, sk, h, hk, d, dtype, causal, mode)
    "c3": (8, 4096, 4096, 32, 32, 128, "bf16", True, "fwd_bwd"),
    "c3_fwd": (8, 4096, 4096, 32, 32, 128, "bf16", True, "fwd"),
    "c2": (4, 2048, 2048, 16, 16, 64, "bf16", True, "fwd"),
    "c5": (128, 1, 8192, 32, 8, 128, "fp16", True, "decode"),
    # flash_attn_varlen_func, ragged batch, GQA 32/8, sliding window (1024, 0) + ALiBi, causal
    "c4": (8, 8192, 8192, 32, 8, 128, "bf16", True, "varlen_fwd"),
}
C4_SEQLENS = [8192, 6144, 4096, 3072, 2048, 1024, 513, 127]
C4_WINDOW = (1024, 0)


def visible_pairs(seqlens_q, seqlens_k, causal, window):
    """Sum over rows of the number of visible keys (the masked FLOP count: 4 h d times this sum)."""
    wl, wr = window
    if causal:
        wr = 0
    tot = 0
    for sq, sk in zip(seqlens_q, seqlens_k):
        for i in range(sq):
            hi = sk if wr < 0 else min(sk, i + 1 + sk - sq + wr)
            lo = 0 if wl < 0 else max(0, i + sk - sq - wl)
            tot += max(0, hi - lo)
    return tot


class Hip:
    """hipEvent access through the HIP runtime already loaded by torch."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so.7") if not hasattr(Hip, "_l") else Hip._l
        Hip._l = self.lib
        self.lib.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.lib.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hipEventSynchronize.argtypes = [C.c_void_p]

    def event(self):
        e = C.c_void_p()
        assert self.lib.hipEventCreate(C.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.lib.hipEventRecord(e, C.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        self.lib.hipEventSynchronize(b)
        assert self.lib.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value


def shard_range(n, world, rank):
    """Contiguous shard [start, stop) of n independent batch entries for `rank` of `world`
    (sizes differ by at most one).  The attention op has no cross-batch dependency, so this is the
    whole multi-GPU decomposition: no collective touches the data path."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def balanced_varlen_split(seqlens_q, seqlens_k, causal, window, world):
    """Cuts a ragged batch into `world` contiguous runs of whole sequences so that the largest
    run's visible query-key pair count (the masked FLOP count up to 4*h*d) is as small as possible
    (exact: dynamic programme over the prefix sums).  Returns [(start, stop)] per rank; a rank may
    get an empty run when there are fewer sequences than ranks."""
    n = len(seqlens_q)
    w = [visible_pairs([a], [b], causal, window) for a, b in zip(seqlens_q, seqlens_k)]
    pre = [0]
    for x in w:
        pre.append(pre[-1] + x)
    INF = float("inf")
    # best[r][i]: minimal max-load cutting the first i sequences into r runs
    best = [[INF] * (n + 1) for _ in range(world + 1)]
    cut = [[0] * (n + 1) for _ in range(world + 1)]
    best[0][0] = 0
    for r in range(1, world + 1):
        for i in range(n + 1):
            for j in range(i + 1):
                c = max(best[r - 1][j], pre[i] - pre[j])
                if c < best[r][i]:
                    best[r][i], cut[r][i] = c, j
    bounds, i = [], n
    for r in range(world, 0, -1):
        j = cut[r][i]
        bounds.append((j, i))
        i = j
    return bounds[::-1]


def launch_ranks(n, argv):
    """Starts n ranks of this script on this node (one per GPU, rendezvous on 127.0.0.1), relays
    rank 0's output and returns the first non-zero exit code.  Called before anything in this
    process has touched a GPU; the children are fresh interpreters, nothing is re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk_:
        sk_.bind(("127.0.0.1", 0))
        port = sk_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # Poll all ranks: the first one that fails ends the launch.  The survivors would otherwise sit
    # in the rendezvous or in a barrier until the process group's time-out.
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0:
                rc = code
                break
    if rc != 0:
        for pr in live:
            pr.terminate()
        deadline = time.time() + 10.0
        for pr in live:
            try:
                pr.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    return rc


def dist_setup(backend):
    """Joins the process group when launched by torch.distributed.run; returns (dist or None, rank,
    world, local_rank).  The group is used only for the timing barrier and the max-over-ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return None, rank, world, local_rank
    import datetime
    import torch.distributed as dist
    # finite: a rank that never arrives fails the others here instead of hanging them
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("VFA_BENCH_PG_TIMEOUT", "300"))))
    return dist, rank, world, local_rank


def max_over_ranks(dist, seconds, device):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def sum_over_ranks(dist, x, device):
    if dist is None:
        return x
    t = torch.tensor([x], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()


def aggregate(work_per_rank_step, steps, world, elapsed_max, scale):
    """Whole-job throughput: the work all ranks did divided by the slowest rank's time."""
    return work_per_rank_step * steps * world / elapsed_max / scale


def pmc_traffic(workload, kernels):
    """HBM bytes per launch of the named kernel(s), from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected with rocprofv3 in runs of their own, never inside a timed run); None
    when no pass exists for this workload."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    try:
        table = json.load(open(found[-1])).get(workload.replace("_fwd", ""), {})
        return sum(table[k]["hbm_bytes"] for k in kernels.split("+")) if table else None
    except (OSError, KeyError, ValueError, IndexError):
        return None


def fwd_flops(b, sq, sk, h, d, causal):
    f = 4.0 * b * h * sq * sk * d
    return f / 2 if causal else f


def build_params(cabi, q, k, v, out, lse, causal, stream, do=None, dq=None, dk=None, dv=None, sd=None,
                 dq_acc=None):
    P = cabi.VfaParams()
    P.dtype = 1 if q.dtype == torch.bfloat16 else 0
    b, sq, h, d = q.shape
    P.q, P.k, P.v, P.out, P.softmax_lse = (t.data_ptr() for t in (q, k, v, out, lse))
    for name, t in (("q", q), ("k", k), ("v", v), ("o", out)):
        cabi._set4(P, name, t)
    P.b, P.seqlen_q, P.seqlen_k, P.h, P.h_k, P.d = b, sq, k.shape[1], h, k.shape[2], d
    P.is_seqlens_k_cumulative = 1
    P.softmax_scale = d ** -0.5
    P.is_causal, P.window_size_left, P.window_size_right = int(causal), -1, -1
    P.num_splits = 1
    P.stream = stream
    if do is not None:
        P.dout, P.dq, P.dk, P.dv = (t.data_ptr() for t in (do, dq, dk, dv))
        for name, t in (("do", do), ("dq", dq), ("dk", dk), ("dv", dv)):
            cabi._set4(P, name, t)
        P.dsoftmax_sum, P.dq_accum = sd.data_ptr(), dq_acc.data_ptr()
    return P


def cpu_baseline(name, b, sq, sk, h, hk, d, dtype, causal, mode):
    """The CPU oracle (a scalar C port of the reference algorithm, OpenMP over (batch, head)) timed
    on a bounded sample of the same workload: batch 1 and as many heads as threads."""
    if mode == "varlen_fwd":
        mode, sq, sk = "fwd", 2048, 2048  # sample: one 2048-token sequence of the ragged batch
    import numpy as np
    from oracle import fa_oracle
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, 16))
    hs = 3 * threads  # three heads per thread: about 10 s of CPU work for the headline shape
    rng = np.random.default_rng(0)
    s_q = sq if mode != "decode" else 1
    if mode == "decode":
        bb, hq, hkv = threads, h, hk
        q = fa_oracle.round_to(rng.standard_normal((bb, 1, hq, d), dtype=np.float32), dtype)
        k = fa_oracle.round_to(rng.standard_normal((bb, sk, hkv, d), dtype=np.float32), dtype)
        v = fa_oracle.round_to(rng.standard_normal((bb, sk, hkv, d), dtype=np.float32), dtype)
        t0 = time.perf_counter()
        fa_oracle.attention_fwd(q, k, v, dtype=dtype, causal=causal,
                                cache_seqlens=np.full(bb, sk, np.int32))
        dt = time.perf_counter() - t0
        gb = 2.0 * bb * sk * hkv * d * 2 / 1e9
        return {"value": gb / dt, "unit": "GB/s", "cores": threads, "kind": "port",
                "sample": f"oracle fwd_kvcache, batch {bb} of {b}, full kv_len {sk}, {dt:.1f} s"}
    q = fa_oracle.round_to(rng.standard_normal((1, s_q, hs, d), dtype=np.float32), dtype)
    k = fa_oracle.round_to(rng.standard_normal((1, sk, hs, d), dtype=np.float32), dtype)
    v = fa_oracle.round_to(rng.standard_normal((1, sk, hs, d), dtype=np.float32), dtype)
    t0 = time.perf_counter()
    out, lse = fa_oracle.attention_fwd(q, k, v, dtype=dtype, causal=causal)
    flops = fwd_flops(1, s_q, sk, hs, d, causal)
    if mode == "fwd_bwd":
        do = fa_oracle.round_to(rng.standard_normal(q.shape, dtype=np.float32), dtype)
        fa_oracle.attention_bwd(do, q, k, v, out, lse, dtype=dtype, causal=causal)
        flops *= 3.5
    dt = time.perf_counter() - t0
    return {"value": flops / dt / 1e12, "unit": "TFLOP/s", "cores": threads, "kind": "port",
            "sample": f"oracle {mode}, batch 1 x {hs} heads of {b} x {h} (full seqlen {sq}), {dt:.1f} s"}


def cpu_sdpa(b, sq, sk, h, d, causal, mode):
    """torch's CPU scaled_dot_product_attention (the other CPU reference the metric names) on the
    same bounded sample shape as cpu_baseline: batch 1, as many heads as threads, full seqlen,
    fp32 and bf16; forward or forward + autograd backward."""
    import torch
    import torch.nn.functional as F
    if mode not in ("fwd", "fwd_bwd"):
        return None
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, 16))
    torch.set_num_threads(threads)
    res = {"cores": threads, "unit": "TFLOP/s",
           "sample": f"torch CPU SDPA {mode}, batch 1 x {threads} heads (full seqlen {sq}), 1 warm-up + best of 2"}
    flops = fwd_flops(1, sq, sk, threads, d, causal) * (3.5 if mode == "fwd_bwd" else 1.0)
    for name, dt_ in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        g = torch.Generator().manual_seed(0)
        q, k, v = (torch.randn(1, threads, n, d, generator=g).to(dt_).requires_grad_(mode == "fwd_bwd")
                   for n in (sq, sk, sk))
        ts = []
        for it in range(3):
            t0 = time.perf_counter()
            o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
            if mode == "fwd_bwd":
                torch.autograd.grad(o, (q, k, v), torch.ones_like(o))
            ts.append(time.perf_counter() - t0)
        res[name] = flops / min(ts[1:]) / 1e12
    return res


GLOBAL_BATCH = {"c5": 1024}   # strong scaling: the decode configuration's 1024 sequences over all ranks


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank runs the per-GPU batch; strong: the global batch is sharded over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sdpa", action="store_true")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1,
                    help="with --selftest-cpu: this rank exits with status 3 before the rendezvous "
                         "(checks that the launcher ends the other ranks and reports the failure)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher / sharding / aggregation self-test on CPU ranks (gloo); the step is a stub, "
                         "no attention work is timed and the line says so")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world == 0 and args.gpus > 1:
        # started plainly: become the launcher.  torch.cuda.device_count() does not initialise the GPU.
        if not args.selftest_cpu and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
        sys.exit(launch_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv)))
    if env_world > 0 and env_world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={env_world}: start one rank per GPU")
    if args.selftest_cpu:
        return selftest_rank(args)
    return run_rank(args)


def selftest_rank(args):
    """One CPU rank of the launcher self-test: the real rendezvous (gloo), sharding, barrier,
    max-over-ranks, work sum and output line of run_rank around a stub step."""
    import numpy as np
    if int(os.environ.get("RANK", "0")) == args.selftest_fail_rank:
        sys.exit(3)
    dist, rank, world, _ = dist_setup("gloo")
    dev = torch.device("cpu")
    b = 8
    lo, hi = shard_range(b, world, rank) if args.scaling == "strong" else (0, b)
    a = np.ones((64, 64), np.float32)
    work = float(hi - lo)

    def step():
        for _ in range(lo, hi):
            a @ a
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(dist, time.perf_counter() - t0, dev)
    total = sum_over_ranks(dist, work, dev)
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (stub step, batch entries per second)",
                          "value": total * args.steps / elapsed, "unit": "entries/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                          "dtype": "f32", "data": "launcher self-test: no attention work",
                          "config": {"workload": "selftest", "entries_all_ranks": total}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args):
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP extension has no CPU fallback)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist, rank, world, local_rank = dist_setup("nccl")  # "nccl" is RCCL on ROCm
    import cabi
    dev_tune = os.environ.get("VFA_BENCH_DEV_TUNE")   # e.g. "3:1,0:1": profile an alternative kernel choice
    if dev_tune:
        # development library + kernel-variant switches (never the headline: the line says so)
        cabi.LIB_PATH = cabi.DEV_LIB_PATH
        cabi._lib = cabi.dev_lib()
        for kv in dev_tune.split(","):
            cabi._lib.vfa_debug_set_tuning(*(int(x) for x in kv.split(":")))
    cabi.lib()
    hip = Hip()

    b, sq, sk, h, hk, d, dtype, causal, mode = WORKLOADS[args.workload]
    strong = args.scaling == "strong"
    c4_lens = C4_SEQLENS
    if strong:
        if mode == "varlen_fwd":
            lo, hi = balanced_varlen_split(C4_SEQLENS, C4_SEQLENS, True, C4_WINDOW, world)[rank]
            c4_lens = C4_SEQLENS[lo:hi]
        else:
            lo, hi = shard_range(GLOBAL_BATCH.get(args.workload, b), world, rank)
            b = hi - lo
        if (mode == "varlen_fwd" and not c4_lens) or b == 0:
            raise SystemExit(f"rank {rank}: empty shard (more ranks than batch entries)")
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(rank)
    stream = torch.cuda.current_stream().cuda_stream
    K, W = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    if mode == "varlen_fwd":
        lens = c4_lens
        total = sum(lens)
        cu = torch.tensor([0] + list(__import__("itertools").accumulate(lens)), dtype=torch.int32, device=dev)
        q = torch.randn(total, h, d, device=dev, dtype=tdt)
        k = torch.randn(total, hk, d, device=dev, dtype=tdt)
        v = torch.randn(total, hk, d, device=dev, dtype=tdt)
        slopes = torch.rand(len(lens), h, device=dev, dtype=torch.float32) * 0.3
        out = torch.empty_like(q)
        lse = torch.empty(len(lens), h, max(lens), device=dev, dtype=torch.float32)
        P = cabi.VfaParams()
        P.dtype = 1
        P.q, P.k, P.v, P.out, P.softmax_lse = (t.data_ptr() for t in (q, k, v, out, lse))
        for name, t in (("q", q), ("k", k), ("v", v), ("o", out)):
            cabi._set4(P, name, t, False)
        P.b, P.seqlen_q, P.seqlen_k, P.h, P.h_k, P.d = len(lens), max(lens), max(lens), h, hk, d
        P.cu_seqlens_q = P.cu_seqlens_k = cu.data_ptr()
        P.is_seqlens_k_cumulative = 1
        P.alibi_slopes, P.alibi_slopes_batch_stride = slopes.data_ptr(), slopes.stride(0)
        P.softmax_scale = d ** -0.5
        P.is_causal, P.window_size_left, P.window_size_right = 1, C4_WINDOW[0], C4_WINDOW[1]
        P.num_splits, P.stream = 1, stream
        step = lambda: cabi.call("vfa_mha_varlen_fwd", P)
        work = 4.0 * h * d * visible_pairs(lens, lens, True, C4_WINDOW)
        unit, metric = "TFLOP/s", "varlen GQA 32/8 d128 window(1024,0)+ALiBi causal fwd TFLOPS (masked FLOPs)"
        kern_name, bound, peak, scale = "fwd_prefill_kernel", "mfma", PEAK_TFLOPS_BF16, 1e12
    elif mode == "decode":
        q = torch.randn(b, 1, h, d, device=dev, dtype=tdt)
        kc = torch.randn(b, sk, hk, d, device=dev, dtype=tdt)
        vc = torch.randn(b, sk, hk, d, device=dev, dtype=tdt)
        cs = torch.full((b,), sk, dtype=torch.int32, device=dev)
        g = h // hk
        # single-token GQA decode: the query heads of a KV group are folded into rows, as the
        # extension module does (flash_api.cpp:747-755)
        qf = q.reshape(b, hk, g, d).transpose(1, 2)
        of = torch.empty(b, hk, g, d, device=dev, dtype=tdt).transpose(1, 2)
        lse = torch.empty(b, hk, g, device=dev, dtype=torch.float32)
        P = build_params(cabi, qf, kc, vc, of, lse, False, stream)
        P.cu_seqlens_k = cs.data_ptr()
        P.is_seqlens_k_cumulative = 0
        step = lambda: cabi.call("vfa_mha_fwd_kvcache", P)
        work = 2.0 * b * sk * hk * d * 2  # K + V bytes streamed once
        unit, metric = "GB/s", "kv-cache decode K+V bytes/s (b128/GPU sq1 kv8192 h32/hk8 d128 fp16)"
        kern_name, bound, peak, scale = "fwd_decode_kernel", "hbm", PEAK_HBM_GBS, 1e9
    else:
        q = torch.randn(b, sq, h, d, device=dev, dtype=tdt)
        k = torch.randn(b, sk, hk, d, device=dev, dtype=tdt)
        v = torch.randn(b, sk, hk, d, device=dev, dtype=tdt)
        out = torch.empty_like(q)
        lse = torch.empty(b, h, sq, device=dev, dtype=torch.float32)
        Pf = build_params(cabi, q, k, v, out, lse, causal, stream)
        f_fwd = fwd_flops(b, sq, sk, h, d, causal)
        if mode == "fwd":
            step = lambda: cabi.call("vfa_mha_fwd", Pf)
            work = f_fwd
            # (d = 128 with two or more 256-row blocks per CU: the persistent-workgroup kernel)
            kern_name = "fwd_prefill_persist_kernel" if d > 64 and not dev_tune else "fwd_prefill_kernel"
        else:
            do = torch.randn_like(q)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            sd = torch.empty(b, h, (sq + 127) // 128 * 128, device=dev, dtype=torch.float32)
            dq_acc = torch.empty(q.shape, device=dev, dtype=torch.float32)
            Pb = build_params(cabi, q, k, v, out, lse, causal, stream, do, dq, dk, dv, sd, dq_acc)

            def step():
                cabi.call("vfa_mha_fwd", Pf)
                cabi.call("vfa_mha_bwd", Pb)
            work = 3.5 * f_fwd
            kern_name = "bwd_dkdv_kernel+bwd_dq_persist_kernel" if not dev_tune else "bwd_dkdv_kernel+bwd_dq_kernel"
        unit = "TFLOP/s"
        metric = ("attention fwd+bwd TFLOPS, seqlen 4096 headdim 128 bf16 causal" if args.workload == "c3"
                  else f"attention {mode} TFLOPS ({args.workload})")
        bound, peak, scale = "mfma", PEAK_TFLOPS_BF16, 1e12

    for _ in range(W):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, elapsed, dev)
    # all ranks' work over the slowest rank's time (shards may differ by one entry)
    value = sum_over_ranks(dist, work, dev) * K / elapsed / scale

    # ---- dominant-kernel duration with HIP events on the launch stream (rank 0, after the timed
    # region so that the region itself carries no instrumentation) ----
    roof = None
    if rank == 0:
        n = max(5, min(K, 20))
        durs = []
        if mode == "fwd_bwd":
            evs = [(C.c_void_p * 4)(*[hip.event() for _ in range(4)]) for _ in range(n)]
            for i in range(n):
                cabi.call("vfa_mha_fwd", Pf)
                Pb.stage_events = C.cast(evs[i], C.c_void_p)
                cabi.call("vfa_mha_bwd", Pb)
            Pb.stage_events = None
            torch.cuda.synchronize()
            # the backward's algorithmic work (2.5 x forward) is done by two kernels: the
            # key-stationary dK/dV kernel (events 1..2) and the query-stationary dQ kernel (2..3)
            t_dkdv = [hip.elapsed_ms(C.c_void_p(e[1]), C.c_void_p(e[2])) for e in evs]
            t_dq = [hip.elapsed_ms(C.c_void_p(e[2]), C.c_void_p(e[3])) for e in evs]
            durs = [a + b_ for a, b_ in zip(t_dkdv, t_dq)]
            kwork = 2.5 * f_fwd
            a1, a2 = sum(t_dkdv) / n, sum(t_dq) / n
            extra["bwd_kernels"] = [
                {"name": "bwd_dkdv_kernel", "avg_ms": a1, "executed_mfma_tflops": 2.0 * f_fwd / a1 / 1e9,
                 "note": "4 MFMA products: S, dP, dV, dK"},
                {"name": kern_name.split("+")[1], "avg_ms": a2, "executed_mfma_tflops": 1.5 * f_fwd / a2 / 1e9,
                 "note": "3 MFMA products: S, dP (recomputed), dQ"}]
            fe = [(hip.event(), hip.event()) for _ in range(n)]
            for a, bb_ in fe:
                hip.record(a, stream); cabi.call("vfa_mha_fwd", Pf); hip.record(bb_, stream)
            torch.cuda.synchronize()
            fd = sorted(hip.elapsed_ms(a, bb_) for a, bb_ in fe)
            extra["fwd_kernel"] = {"name": "fwd_prefill_kernel" if dev_tune else "fwd_prefill_persist_kernel",
                                   "avg_ms": sum(fd) / len(fd),
                                   "tflops": f_fwd / (sum(fd) / len(fd)) / 1e9}
        else:
            fe = [(hip.event(), hip.event()) for _ in range(n)]
            for a, bb_ in fe:
                hip.record(a, stream); step(); hip.record(bb_, stream)
            torch.cuda.synchronize()
            durs = [hip.elapsed_ms(a, bb_) for a, bb_ in fe]
            kwork = work
        avg = sum(durs) / len(durs)
        achieved = kwork / (avg * 1e-3) / scale
        roof = {"bound": bound, "kernel": kern_name, "achieved": achieved, "peak": peak,
                "unit": unit, "frac": achieved / peak, "avg_kernel_ms": avg,
                "traffic": pmc_traffic(args.workload, kern_name)}

    # ---- torch SDPA on the same GPU and inputs (rank 0) ----
    if rank == 0 and not args.no_sdpa and mode in ("fwd", "fwd_bwd"):
        import torch.nn.functional as F
        qt, kt, vt = (x.transpose(1, 2).detach().requires_grad_(mode == "fwd_bwd") for x in (q, k, v))
        gt = torch.randn_like(qt)

        def sdpa_step():
            o = F.scaled_dot_product_attention(qt, kt, vt, is_causal=causal)
            if mode == "fwd_bwd":
                o.backward(gt)
                qt.grad = kt.grad = vt.grad = None
        try:
            for _ in range(3):
                sdpa_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ns = max(3, min(K, 10))
            for _ in range(ns):
                sdpa_step()
            torch.cuda.synchronize()
            st = (time.perf_counter() - t1) / ns
            extra["sdpa_gpu"] = {"tflops": work / st / 1e12, "speedup_vs_sdpa": (work * K / elapsed / 1e12) / (work / st / 1e12) / world}
        except Exception as e:  # noqa: BLE001
            extra["sdpa_gpu"] = {"error": str(e)[:100]}

    if rank == 0:
        line = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: b{b} sq{sq} sk{sk} h{h}/{hk} d{d} {dtype} causal={causal} {mode}",
                       "per_gpu_batch": len(c4_lens) if mode == "varlen_fwd" else b,
                       "parallelism": f"batch-sharded replicas x{world}, no collective",
                       **({"dev_tuning": dev_tune} if dev_tune else {})},
            "roofline": roof,
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, b, sq, sk, h, hk, d, dtype, causal, mode)
            sd = cpu_sdpa(b, sq, sk, h, d, causal, mode)
            if sd:
                line["cpu_sdpa"] = sd
        line.update(extra)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
