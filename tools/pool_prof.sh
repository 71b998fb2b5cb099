#!/bin/bash
# kernel times of the scoring tail (tools/tail_bench.py) under rocprofv3 --kernel-trace --stats: the pooling kernels' average
# duration per launch.   usage: tools/pool_prof.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/tail_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/${tag}_tail_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_tail_stats -o p -- python3 $R/tools/tail_bench.py > /dev/null 2>&1
s=$(find $O/${tag}_tail_stats -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s $O/${tag}_tail_kernel_stats.csv && grep -E "pool|minmax|grp_|ref_score|ref_iou|coherence" $s | cut -d, -f1-4 | cut -c1-160
find $O/${tag}_tail_stats -name "*kernel_trace.csv" -delete 2>/dev/null
