#!/bin/bash
# Counter passes (rocprofv3 --pmc, one pass per counter group) over any command of this repo.
#   tools/pmc_run.sh <tag> <kernel-name pattern> <python script> [args...]   -> gpurun_out/pmc_<tag>/summary.json
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pass$i -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $out/pass$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_collect.py $out "$pat" > $out/summary.json
cat $out/summary.json
