#!/bin/bash
# The driver's headline command repeated, without and with an rocm-smi poller beside it (the driver samples rocm-smi every
# ~5 s during its run): how much does the 0.6 s timed region move from run to run?
#   usage: tools/headline_repeat.sh <tag> [runs]
tag=$1; runs=${2:-4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
F="--no-also --no-live-pmc --no-cpu-baseline --no-rccl-check --no-disk"
one() { python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['value'],2), round(d['ms_per_step'],2), d['timed_region'])"; }
for i in $(seq $runs); do one plain; done > $O/${tag}_repeat.log
( while true; do rocm-smi --showuse --showmemuse --showpower --showclocks --json > /dev/null 2>&1; sleep 0.5; done ) &
poller=$!
for i in $(seq $runs); do one smi_poller; done >> $O/${tag}_repeat.log
kill $poller
cat $O/${tag}_repeat.log
