# needs the diagnostic twin (make -C hybridgl_amd/csrc diag): the knock-outs are not compiled into the product library
cd /tmp && export TMPDIR=/tmp
export HGL_LIB_NAME=libhybridgl_diag.so
for D in 0 1 2 3 4 8 16 31; do
  HGL_ATTN_PS_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg_$D -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_win_one.py 16 > /dev/null 2>&1
  s=$(find $GRAFT_REPO_ROOT/gpurun_out/dbg_$D -name "*kernel_stats.csv" | head -1)
  echo "dbg=$D $(grep -E 'attn_psp' $s | cut -d, -f2-4)"
done
find $GRAFT_REPO_ROOT/gpurun_out/ -name "*kernel_trace.csv" -delete
