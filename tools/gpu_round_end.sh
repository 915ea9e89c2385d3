#!/bin/bash
# Everything the round's records come from, on one GPU box: tests, bench lines, kernel-trace
# stats, the FETCH_SIZE / WRITE_SIZE counter passes and the instruction-mix passes.  Each GPU step
# has its own time limit and the chain stops at the first failure.
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
timeout -k 10 300 python tools/perf_headdims.py 160 192 256 > $O/headdims.txt 2>&1
timeout -k 10 600 python tools/seqlen_sweep.py $O/seqlen_sweep.json > $O/seqlen_sweep.log 2>&1
echo sweeps done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-sdpa"
for w in c3 c5 c2 c4; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o $w -- python3 bench.py --workload $w $B > $O/trace_$w.log 2>&1
done
echo traces done
S="--steps 3 --warmup 1 --no-cpu-baseline --no-sdpa"
for w in c3 c5 c2 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$w/$c -o p -- python3 bench.py --workload $w $S > $O/pmc_${w}_$c.log 2>&1
  done
done
# matrix-pipe busy share of the headline kernels (one pass: 2 SQ + 1 GRBM counter)
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3/MFMA -o p -- python3 bench.py $S > $O/pmc_c3_MFMA.log 2>&1
echo traffic done
python tools/pmc_summarize.py $O/pmc_traffic.json c3=$O/pmc_c3 c5=$O/pmc_c5 c2=$O/pmc_c2 c4=$O/pmc_c4
# instruction mix of the headline kernels: three passes of <= 8 SQ counters, for the shipped
# dispatch ("persistent": forward and dQ walk blocks) and for one block per workgroup ("block":
# development library, switch 3 = 1)
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P3="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $O/issue_persistent/p$i -o p -- python3 bench.py $S > $O/issue_persistent_p$i.log 2>&1
  VFA_BENCH_DEV_TUNE=3:1 timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $O/issue_block/p$i -o p -- python3 bench.py $S > $O/issue_block_p$i.log 2>&1
done
python tools/pmc_issue.py $O/pmc_issue.json persistent=$O/issue_persistent block=$O/issue_block
echo pmc done
find $O -name "*kernel_stats.csv" | head
# keep the merged-back payload small
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
