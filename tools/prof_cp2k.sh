cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cp2k -o cp -- python3 $GRAFT_REPO_ROOT/tools/bench_cp2k.py 524288 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_cp2k.log 2>&1
