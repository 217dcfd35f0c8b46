#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command on the final build (without the CPU baseline and the
# concurrent-streams leg: tracing crashes inside hipGraphLaunch with 16 streams)   -> gpurun_out/r4final_stats/
O=$GRAFT_REPO_ROOT/gpurun_out/r4final_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-streams-leg > $O/bench.json 2> $O/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/stats -name "bench_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-170
rm -rf $O/stats
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('value', d['value'], 'us/launch', r['us_per_launch'], 'frac', r['frac'])"
