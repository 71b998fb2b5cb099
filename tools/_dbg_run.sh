cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_sam.py -x -q -k "decoder" 2>&1 | tail -3
for kern in auto P; do
export HGL_X3_KERNEL=$kern
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o out -- python3 tools/decoder_bench.py 6 8 2>&1 | grep "decoder,"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/pd/out_kernel_stats.csv")))
for r in rows[:40]:
    if any(k in r['Name'] for k in ('gemm_x3p','gemm_f16x3_kernel','combine')):
        print(r['Name'][:90].ljust(90), r['Calls'], round(int(r['TotalDurationNs'])/9/1e3,1), round(float(r['AverageNs'])/1e3,1))
PY
done
