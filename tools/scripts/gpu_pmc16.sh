cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "FETCH_SIZE WRITE_SIZE"; do
  tag=$(echo $set | cut -c1-10 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc16_$tag -o p --output-format csv -- python $R/tools/conv_bench.py --shape s0,s1,s3,s2 --impl 4 --iters 10 > $R/gpurun_out/pmc16_$tag.log 2>&1
done
ls $R/gpurun_out | grep pmc16
