set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/pmc_c2x; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
S="--steps 3 --warmup 1 --no-cpu-baseline --no-sdpa"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P3="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $O/c2/p$i -o p -- python3 bench.py --workload c2 $S > $O/c2_p$i.log 2>&1
done
python tools/pmc_issue.py $O/pmc_issue_c2.json c2=$O/c2
find $O -name "*counter_collection.csv" -size +5M -delete
