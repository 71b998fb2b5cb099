#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_clip.py -m gpu -x -q -k "blur" 2>&1 | tail -2
python - <<'PY'
import torch, time
from hybridgl_amd import ops
img = torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(5): ops.gaussian_blur_u8(img, 15)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.gaussian_blur_u8(img, 15)
e1.record(); torch.cuda.synchronize()
print("blur 480x640x3 k15: %.1f us per call" % (e0.elapsed_time(e1) * 5))
PY
cat > /tmp/blur_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from hybridgl_amd import ops
img = torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(20): ops.gaussian_blur_u8(img, 15)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/blurq -o p -- python3 /tmp/blur_one.py > /dev/null 2>&1
python tools/stats_top.py gpurun_out/blurq 1 4
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/blurq_$c -o p -- python3 /tmp/blur_one.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/blurq_$c/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "blur" in r["Kernel_Name"]]
print("$c per launch (raw counter units):", sum(v) / 20)
PY
done
