#!/bin/bash
# PhraseCut-shaped items, per-kernel table (serial stages).  usage: pc_quick.sh <tag>
tag=${1:-pq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_clip.py -m gpu -x -q -k "blur" 2>&1 | tail -2
python tools/phrasecut_profile.py 4 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_pc -o p --output-format csv -- python3 tools/phrasecut_profile.py 4 > /dev/null 2>&1
python tools/stats_top.py gpurun_out/${tag}_pc 6 40
