# functional check of the N>1 launch contract on a 1-GPU box: both ranks map onto device 0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/multi
mkdir -p $O
python -m pytest tests -m gpu -q -x -k "graph_replay" 2>&1 | tail -40
for n in 2; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 2000 --warmup 200 > $O/bench_n$n.json 2> $O/bench_n$n.err
echo "rc=$? lines=$(wc -l < $O/bench_n$n.json)"; python -c "
import json; d=json.load(open('$O/bench_n$n.json')); print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}, d['config']['parallelism'], 'cpu_baseline' in d, d['roofline']['frac'])"
done
