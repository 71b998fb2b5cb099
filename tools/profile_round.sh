#!/bin/bash
# Evidence pass of a round on the GPU box: bench line, rocprofv3 kernel stats of the SAME bench command, PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE in separate runs), SQ counters of the dominant GEMM, the two-rank bench, a sustained run
# with clock samples.    usage: tools/profile_round.sh <tag>        -> gpurun_out/<tag>_*
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
tail -c 600 $O/${tag}_bench_default.json | head -c 300; echo
python3 $R/bench.py --gpus 2 --backend gloo --steps 32 --warmup 8 --no-cpu-baseline --no-also > $O/${tag}_bench_2ranks_gloo.json 2> $O/${tag}_bench_2ranks_gloo.err
head -c 260 $O/${tag}_bench_2ranks_gloo.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o p -- python3 $R/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-also > $O/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_serial -o p -- python3 $R/tools/group_profile.py 2 16 > $O/${tag}_stats_serial.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_fetch -o p -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-also > $O/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_write -o p -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-also > $O/${tag}_write.log 2>&1
# sustained run + clock samples (rocm-smi is read-only here)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/${tag}_sustained_clocks.txt &
SMI=$!
python3 $R/bench.py --steps 400 --warmup 16 --no-cpu-baseline --no-also > $O/${tag}_bench_sustained.json 2> $O/${tag}_bench_sustained.err
kill $SMI 2>/dev/null
head -c 260 $O/${tag}_bench_sustained.json; echo
bash $R/tools/pmc_gemm.sh ${tag}_x3p_clipqkv 201728 2304 768 P > /dev/null 2>&1
cat $O/pmc_${tag}_x3p_clipqkv/summary.json | head -40
# keep the summaries, drop the per-dispatch traces (tens of MB)
find $O/${tag}_stats $O/${tag}_stats_serial -name "*kernel_trace.csv" -delete 2>/dev/null
python3 $R/tools/profile_summary.py ${tag} $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write > $O/${tag}_traffic.log 2>&1
cp $R/profiles/${tag}_pmc_traffic.json $R/profiles/${tag}_kernel_stats.csv $O/ 2>/dev/null
rm -rf $O/${tag}_fetch $O/${tag}_write
rm -rf $O/pmc_${tag}_x3p_clipqkv/pass*
