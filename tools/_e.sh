cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests/test_gpu_sam.py $R/tests/test_gpu_primitives.py $R/tests/test_gpu_attention_ps.py -x -q -m gpu 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs -o p -- python3 $R/tools/group_profile.py 2 16 > /dev/null 2>&1
python3 $R/tools/stats_top.py /tmp/gs 48 80 | grep -E "total|relpos|attn_psp_kernel<80, 2"
