"""world_size-2 gloo test (CPU) of the multi-GPU decomposition bench.py uses: batch entries are
sharded contiguously over ranks with no data-path collective; the only collectives are the timing
barrier and the max-over-ranks.  Each rank runs the CPU oracle on its shard; the gathered shards
must equal the unsharded result bit for bit (batch entries are independent)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import bench
    from oracle import fa_oracle
    d, r, w, lr = bench.dist_setup("gloo")
    assert (r, w, lr) == (rank, world, rank)
    rng = np.random.default_rng(0)  # same data on every rank; each takes its shard
    b, s, h, hd = 5, 48, 2, 64
    q, k, v = (rng.standard_normal((b, s, h, hd), dtype=np.float32) for _ in range(3))
    lo, hi = bench.shard_range(b, w, r)
    out, lse = fa_oracle.attention_fwd(q[lo:hi], k[lo:hi], v[lo:hi], dtype="bf16", causal=True)
    gathered = [None] * w
    d.all_gather_object(gathered, (lo, hi, out))
    tmax = bench.max_over_ranks(d, 1.0 + r, torch.device("cpu"))
    assert tmax == float(w)                      # slowest rank's time
    assert bench.aggregate(10.0, 4, w, tmax, 1.0) == 10.0 * 4 * w / w
    if r == 0:
        full, _ = fa_oracle.attention_fwd(q, k, v, dtype="bf16", causal=True)
        cat = np.concatenate([g[2] for g in sorted(gathered)], axis=0)
        assert [g[:2] for g in sorted(gathered)][0][0] == 0 and sorted(gathered)[-1][1] == b
        ret["equal"] = bool(np.array_equal(cat, full))
    d.barrier()
    d.destroy_process_group()


def test_shard_range_covers_and_balances():
    sys.path.insert(0, ROOT)
    import bench
    for n in (1, 7, 8, 128, 1024, 1025):
        for w in (1, 2, 3, 8):
            parts = [bench.shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_range(1024, 8, 3) == (384, 512)   # decode configuration: 128 sequences per GPU


def test_two_ranks_gloo_sharded_equals_unsharded():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("equal") is True


def _run_bench(*argv, env=None):
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` started plainly (no WORLD_SIZE) must run TWO ranks: the launcher,
    the gloo rendezvous, the shard of each rank, the max / sum over ranks and the one JSON line of
    rank 0 are exercised here on CPU ranks around a stub step (--selftest-cpu); the GPU ranks run
    the same code with the attention step in its place."""
    import json
    for scaling, entries in (("weak", 16.0), ("strong", 8.0)):
        r = _run_bench("--gpus", "2", "--selftest-cpu", "--steps", "3", "--warmup", "1", "--scaling", scaling)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout            # rank 0 only
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["steps"] == 3
        assert line["config"]["entries_all_ranks"] == entries   # weak: 8 per rank; strong: 8 in all
        assert line["value"] > 0


def test_launcher_fails_fast_when_a_rank_dies_before_the_rendezvous():
    """A rank that exits before joining the process group must end the launch at once with its
    exit status: the other ranks are terminated instead of waiting in the rendezvous for the
    process group's time-out."""
    import time
    for bad in (1, 0):
        t0 = time.time()
        r = _run_bench("--gpus", "2", "--selftest-cpu", "--selftest-fail-rank", str(bad), "--steps", "2", "--warmup", "0")
        assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
        assert time.time() - t0 < 60, "launcher waited for the surviving rank's time-out"


def test_bench_gpus_flag_fails_cleanly_without_the_gpus():
    """Asking for more GPUs than the node has is an error, not a silent one-rank run; so is a
    --gpus that contradicts the launcher's WORLD_SIZE."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("node has the GPUs")
    r = _run_bench("--gpus", "2")
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    r = _run_bench("--gpus", "2", "--selftest-cpu", env={"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_balanced_varlen_split():
    """Ragged batches are cut at sequence boundaries so that the most loaded rank has as few
    visible query-key pairs as a contiguous cut allows."""
    sys.path.insert(0, ROOT)
    import bench
    lens = bench.C4_SEQLENS
    load = [bench.visible_pairs([n], [n], True, bench.C4_WINDOW) for n in lens]
    for w in (1, 2, 4, 8):
        parts = bench.balanced_varlen_split(lens, lens, True, bench.C4_WINDOW, w)
        assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == len(lens)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        worst = max(sum(load[a:b]) for a, b in parts)
        # brute force over all contiguous cuts
        import itertools
        best = min(max(sum(load[a:b]) for a, b in zip((0,) + c, c + (len(lens),)))
                   for c in itertools.combinations_with_replacement(range(len(lens) + 1), w - 1))
        assert worst == best
    # two ranks: the 8192-token sequence alone is NOT half of the windowed work
    assert bench.balanced_varlen_split(lens, lens, True, bench.C4_WINDOW, 2)[0][1] >= 2
