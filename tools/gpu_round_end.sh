#!/bin/bash
# Everything the round's records come from, on one GPU box: tests, bench lines, kernel-trace
# stats and the FETCH_SIZE / WRITE_SIZE counter passes.  Each GPU step has its own time limit and
# the chain stops at the first failure.
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/final   # copy what is to be kept into profiles/rNN_*
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
tail -2 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for w in c3_fwd c2 c5 c4; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
done
echo benches done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -o c3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sdpa > $O/trace_c3.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o c5 -- python3 bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline --no-sdpa > $O/trace_c5.log 2>&1
echo traces done
for w in c3 c5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$w/$c -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa > $O/pmc_${w}_$c.log 2>&1
  done
done
# matrix-pipe busy share of the headline kernels (one pass: 2 SQ + 1 GRBM counter)
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3/MFMA -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sdpa > $O/pmc_c3_MFMA.log 2>&1
echo pmc done
python tools/pmc_summarize.py $O/pmc_traffic.json c3=$O/pmc_c3 c5=$O/pmc_c5
find $O -name "*kernel_stats.csv" | head
# keep the merged-back payload small
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
