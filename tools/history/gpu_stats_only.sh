# only the kernel-trace/stats pass of tools/gpu_profiles.sh
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r1
mkdir -p $O; rm -rf $O/stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-streams-leg --also limits,floating,obstacles,cfg3,cfg4,cfg5 > $O/stats.log 2>&1
tail -2 $O/stats.log | cut -c1-300; ls $O/stats
