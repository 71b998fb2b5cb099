#!/bin/bash
# On ONE box: the driver's command with / without the secondary legs, with / without prepare + balanced groups; prints the
# headline and the stage timeline of each.   usage: tools/headline_ab.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
F="--no-also --no-live-pmc --no-cpu-baseline --no-rccl-check --no-disk"
show() { python3 -c "
import sys,json
d=json.loads(open('$1').readline()); t=d['timed_region']
print('$2', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms/step; enqueue', round(t['host_enqueue_seconds'],3), 'mallocs', t['device_mallocs'], 'prepared', d['config'].get('prepared'))
for s in t['stages']: print('    ', s)
"; }
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $F --no-prepare --no-balance > $O/${tag}_1_reduced_old.json 2>/dev/null; show $O/${tag}_1_reduced_old.json reduced_old
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-prepare --no-balance > $O/${tag}_2_full_old.json 2>/dev/null; show $O/${tag}_2_full_old.json full_old
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_3_full_new.json 2>$O/${tag}_3_full_new.err; show $O/${tag}_3_full_new.json full_new
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $F > $O/${tag}_4_reduced_new.json 2>/dev/null; show $O/${tag}_4_reduced_new.json reduced_new
