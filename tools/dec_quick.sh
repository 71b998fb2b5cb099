#!/bin/bash
# quick decoder loop: the decoder tests, the 529-prompt timing, and a per-kernel table.  usage: dec_quick.sh <tag>
tag=${1:-dq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_sam.py -m gpu -x -q 2>&1 | tail -3
python tools/decoder_bench.py 10 23 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_dec -o p --output-format csv -- python3 tools/decoder_bench.py 5 23 > /dev/null 2>&1

python tools/stats_top.py gpurun_out/${tag}_dec 1 14
