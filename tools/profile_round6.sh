#!/bin/bash
# Round-6 evidence pass on the GPU box (ends with the DRIVER'S command, python3 bench.py --gpus 1 --steps 20 --warmup 5): bench line (with roofline.hbm_kernels and the PhraseCut decoder sub-object), kernel
# stats of the serial group and of the overlapped loop, PMC traffic, SQ counters of the three pre-split attention kernels,
# the decoder's HBM bytes per prompt, the evaluator from disk as 8 ranks / 1 rank.
#   usage: tools/profile_round6.sh <tag>        -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# FIRST the mask decoder, 529 prompts per call (3 warm-up + 5 timed calls): bytes per prompt -- the bench line below replays the
# newest profiles/r*_decoder_traffic.json, which is then this one
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_dec_fetch -o p -- python3 $R/tools/decoder_bench.py 5 23 > $O/${tag}_dec_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_dec_write -o p -- python3 $R/tools/decoder_bench.py 5 23 > $O/${tag}_dec_write.log 2>&1
python3 $R/tools/decoder_traffic.py $O/${tag}_dec_fetch $O/${tag}_dec_write 8 529 ${tag} > $O/${tag}_decoder_traffic.log 2>&1
cp $R/profiles/${tag}_decoder_traffic.json $O/ 2>/dev/null
python3 $R/bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
tail -c 300 $O/${tag}_bench_default.json | head -c 200; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_serial -o p -- python3 $R/tools/group_profile.py 3 16 > $O/${tag}_stats_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o p -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-also --no-disk --no-rccl-check --no-live-pmc > $O/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_fetch -o p -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-also --no-disk --no-rccl-check --no-live-pmc > $O/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_write -o p -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-also --no-disk --no-rccl-check --no-live-pmc > $O/${tag}_write.log 2>&1
for f in $O/${tag}_stats_serial; do
  s=$(find $f -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $O/${tag}_serial_group_kernel_stats.csv
done
python3 $R/tools/profile_summary.py ${tag} $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write > $O/${tag}_traffic.log 2>&1
cp $R/profiles/${tag}_pmc_traffic.json $R/profiles/${tag}_kernel_stats.csv $O/ 2>/dev/null
# the pre-split attention kernels: SAM windows + global blocks (16 images), CLIP (1024 x 12 x 197 x 64)
bash $R/tools/pmc_run.sh ${tag}_attn_win "attn_psp_kernel<80, 1, 1>" tools/attn_win_one.py 16 > /dev/null 2>&1
bash $R/tools/pmc_run.sh ${tag}_attn_glob "attn_psp_kernel<80, 2, 1>" tools/attn_win_one.py 16 > /dev/null 2>&1
bash $R/tools/pmc_run.sh ${tag}_attn_clip "attn_ps_kernel<64, 0, 2>" tools/attn_ps_bench.py > /dev/null 2>&1
# the residual GEMMs beside their twins without the residual, and the row-balanced launch (whole rounds + split-K tail) A/B
for s in group16 group10; do for b in 0 1; do echo "== $s balanced=$b"; X3_SHAPES=$s X3_BALANCED=$b python3 $R/tools/x3_bench.py 2>&1 | grep -v amdgpu.ids; done; done > $O/${tag}_x3_balanced_ab.log
# the scoring tail: pooling kernel time per launch (tools/pool_prof.sh)
bash $R/tools/pool_prof.sh ${tag} > /dev/null 2>&1
python3 $R/tools/decoder_bench.py 10 23 > $O/${tag}_decoder_bench.log 2>&1
# PhraseCut-shaped items, stages back to back on one stream: per-kernel table of the heavy-AMG path
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_pc -o p -- python3 $R/tools/phrasecut_profile.py 4 > $O/${tag}_phrasecut_profile.log 2>&1
s=$(find $O/${tag}_pc -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $O/${tag}_phrasecut_kernel_stats.csv
rm -rf $O/${tag}_pc
python3 $R/tools/phrasecut_profile.py 4 >> $O/${tag}_phrasecut_profile.log 2>&1
python3 $R/tools/tail_bench.py > $O/${tag}_tail_bench.log 2>&1
python3 $R/tools/attn_ps_ab.py 16 > $O/${tag}_attn_ps_ab.log 2>&1
# the evaluator fed from disk as 8 ranks on this one GPU (832 images: 104 per rank) and as 1 rank
python3 $R/tools/evaluator_ranks.py --ranks 8 --images 832 --group 8 > $O/${tag}_ranks8.log 2>&1
tail -1 $O/${tag}_ranks8.log > $O/${tag}_evaluator_8ranks_gloo.json
python3 $R/tools/evaluator_ranks.py --ranks 1 --images 832 --group 16 > $O/${tag}_ranks1.log 2>&1
tail -1 $O/${tag}_ranks1.log > $O/${tag}_evaluator_1rank.json
for t in attn_win attn_glob attn_clip; do cp $O/pmc_${tag}_$t/summary.json $O/${tag}_sq_counters_$t.json 2>/dev/null; rm -rf $O/pmc_${tag}_$t/pass*; done
find $O/${tag}_stats $O/${tag}_stats_serial -name "*kernel_trace.csv" -delete 2>/dev/null
rm -rf $O/${tag}_fetch $O/${tag}_write $O/${tag}_dec_fetch $O/${tag}_dec_write
# LAST: the driver's own command, as the driver runs it (a fresh process after everything above), and its repeatability
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench_driver_cmd.json 2> $O/${tag}_bench_driver_cmd.err
bash $R/tools/headline_repeat.sh ${tag} 3 > /dev/null 2>&1
python3 -c "
import json
d=json.loads(open('$O/${tag}_bench_driver_cmd.json').readline())
print('driver cmd:', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms/step', d['timed_region'])
print({k:round(v['ms_per_step'],2) for k,v in d['also'].items() if 'ms_per_step' in v})
"
cat $O/${tag}_repeat.log
ls -la $O | grep ${tag}_ | head -60
