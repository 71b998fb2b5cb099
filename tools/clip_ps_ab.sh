cd /tmp && export TMPDIR=/tmp
for V in 2 0 2 0; do
  HGL_ATTN_PS_CLIPBLOCKS=$V rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/clipab_$V -o p -- python3 $GRAFT_REPO_ROOT/tools/group_profile.py 3 16 > $GRAFT_REPO_ROOT/gpurun_out/clipab_$V.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/clipab_$V.log
  python3 $GRAFT_REPO_ROOT/tools/stats_top.py $GRAFT_REPO_ROOT/gpurun_out/clipab_$V 64 60 | grep -E "total|attn_|gemm_x3p_kernel<0, 0|layernorm_split_kernelILi3"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/clipab_$V
done
