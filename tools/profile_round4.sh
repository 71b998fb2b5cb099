#!/bin/bash
# Round-4 evidence pass on the GPU box (tools/profile_round.sh + what VERDICT r03 asked for): bench line with the
# evaluator-from-disk leg, kernel stats of the serial group, PMC traffic, SQ counters of the ping-pong GEMM in its NT = 3
# and NT = 2 (fp16-valued weights) instantiations, SQ counters of the window attention, the 8-rank evaluator from disk.
#   usage: tools/profile_round4.sh <tag>        -> gpurun_out/<tag>_*
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
tail -c 400 $O/${tag}_bench_default.json | head -c 200; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_serial -o p -- python3 $R/tools/group_profile.py 3 16 > $O/${tag}_stats_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o p -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-also --no-disk --no-rccl-check --no-live-pmc > $O/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_fetch -o p -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-also --no-disk --no-rccl-check --no-live-pmc > $O/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_write -o p -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-also --no-disk --no-rccl-check --no-live-pmc > $O/${tag}_write.log 2>&1
for f in $O/${tag}_stats_serial; do
  s=$(find $f -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $O/${tag}_serial_group_kernel_stats.csv
done
python3 $R/tools/profile_summary.py ${tag} $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write > $O/${tag}_traffic.log 2>&1
cp $R/profiles/${tag}_pmc_traffic.json $R/profiles/${tag}_kernel_stats.csv $O/ 2>/dev/null
# the ping-pong GEMM on CLIP qkv of a group of 16: genuine fp32 weights (NT = 3) and fp16-valued weights (NT = 2)
bash $R/tools/pmc_gemm.sh ${tag}_x3p_nt3 201728 2304 768 P > /dev/null 2>&1
X3_FP16_VALUED_W=1 bash $R/tools/pmc_gemm.sh ${tag}_x3p_nt2 201728 2304 768 P > /dev/null 2>&1
# SAM's attention kernels (16 images: 6400 window items, 256 global items per block)
bash $R/tools/pmc_run.sh ${tag}_attn_win "attn_x3_kernel<80, 14, 8>" tools/attn_win_one.py 16 > /dev/null 2>&1
bash $R/tools/pmc_run.sh ${tag}_attn_glob "attn_x3_kernel<80, 0, 4>" tools/attn_win_one.py 16 > /dev/null 2>&1
# the evaluator fed from disk as 8 ranks on this one GPU (832 images: 104 per rank) and as 1 rank
python3 $R/tools/evaluator_ranks.py --ranks 8 --images 832 --group 8 > $O/${tag}_ranks8.log 2>&1
tail -1 $O/${tag}_ranks8.log > $O/${tag}_evaluator_8ranks_gloo.json
python3 $R/tools/evaluator_ranks.py --ranks 1 --images 832 --group 16 > $O/${tag}_ranks1.log 2>&1
tail -1 $O/${tag}_ranks1.log > $O/${tag}_evaluator_1rank.json
for t in x3p_nt3 x3p_nt2 attn_win attn_glob; do cp $O/pmc_${tag}_$t/summary.json $O/${tag}_sq_counters_$t.json 2>/dev/null; rm -rf $O/pmc_${tag}_$t/pass*; done
find $O/${tag}_stats $O/${tag}_stats_serial -name "*kernel_trace.csv" -delete 2>/dev/null
rm -rf $O/${tag}_fetch $O/${tag}_write
ls -la $O | grep ${tag}_ | head -40
