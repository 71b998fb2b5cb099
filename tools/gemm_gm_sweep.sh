#!/bin/bash
# does the ping-pong GEMM's clock move with its L2 traffic?  Sweeps the L2 blocking (HGL_X3_GM = row
# tiles per tile group) on two pipeline shapes and records, per setting, launch time, FETCH_SIZE, the clock held
# (GRBM_GUI_ACTIVE / 8 / time) and the matrix pipe's busy share -- one table.
#   usage: tools/gemm_gm_sweep.sh <tag>        -> gpurun_out/<tag>_gm_sweep.txt
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export X3_KERNEL=P
export HGL_LIB_NAME=libhybridgl_diag.so     # make -C hybridgl_amd/csrc diag: HGL_X3_GM is a diagnostic switch
out=$O/${tag}_gm_sweep.txt
: > $out
for shape in "201728 2304 768" "65536 5120 1280" "65536 3840 1280"; do
  for gm in 1 2 3 4 6 8 12 16 32; do
    d=$O/gmsweep_${gm}
    rm -rf $d; mkdir -p $d
    HGL_X3_GM=$gm timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $d/a -o p -- python3 $R/tools/x3_one.py $shape 4 > $d/a.log 2>&1
    HGL_X3_GM=$gm timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d/b -o p -- python3 $R/tools/x3_one.py $shape 4 > $d/b.log 2>&1
    HGL_X3_GM=$gm timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d/c -o p -- python3 $R/tools/x3_one.py $shape 6 > $d/c.log 2>&1
    python3 - "$shape" $gm $d >> $out <<'PY'
import csv, glob, sys
shape, gm, d = sys.argv[1], sys.argv[2], sys.argv[3]
def counters(sub):
    acc = {}
    for p in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "gemm_x3p" in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}
def dur(sub):
    t = []
    for p in glob.glob(f"{d}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "gemm_x3p" in r["Kernel_Name"]:
                t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    t = t[1:] if len(t) > 1 else t
    return sum(t) / max(len(t), 1)
a, b = counters("a"), counters("b")
ta, tc = dur("a"), dur("c")
M, N, K = (int(v) for v in shape.split())
fl = 2.0 * M * N * K
clock = a.get("GRBM_GUI_ACTIVE", 0) / 8 / max(ta, 1e-9) / 1e3
busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(a.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
print(f"{shape:>18s} gm {gm:>2s}: unprofiled {tc:8.1f} us = {fl / tc / 1e6:6.1f} TF/s | profiled {ta:8.1f} us, clock {clock:5.3f} GHz, matrix pipe busy {busy:5.3f} | FETCH_SIZE x2 {b.get('FETCH_SIZE', 0) * 2 * 1024 / 1e9:6.3f} GB (operands {(M * K + N * K) * 4 / 1e9:5.3f} GB)")
PY
    rm -rf $d
  done
done
cat $out
