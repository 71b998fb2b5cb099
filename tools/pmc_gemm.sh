#!/bin/bash
# Counter passes (rocprofv3 --pmc, one pass per counter group) over one f16x3 GEMM shape.
#   tools/pmc_gemm.sh <tag> <M> <N> <K> [kernel-kind]      -> gpurun_out/pmc_<tag>/*.json
tag=$1; M=$2; N=$3; K=$4; kind=${5:-auto}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
export X3_KERNEL=$kind
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pass$i -o p -- python3 $GRAFT_REPO_ROOT/tools/x3_one.py $M $N $K 4 > $out/pass$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_collect.py $out gemm_x3 > $out/summary.json
cat $out/summary.json
