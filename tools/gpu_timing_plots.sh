#!/bin/bash
# The reference's published benchmark (examples/timing_plots.py) on this package and on the staged reference, same box.
# Run through gpurun from the repo root; results under gpurun_out/tp/ (copy timing_plots.json into profiles/round3/).
set -u
mkdir -p gpurun_out/tp
python tests/timing_plots_replica.py gpurun_out/tp/timing_plots.json > gpurun_out/tp/log.txt 2>&1
echo "replica rc=$?"; tail -12 gpurun_out/tp/log.txt
# the six-row line with the training signal among the outputs, under its own kernel name
ABRK_BENCH_TS=1 python bench.py --workload osc6 --no-extras --sustain-seconds 2 > gpurun_out/tp/bench_osc6_ts.json 2> gpurun_out/tp/bench_osc6_ts.err
echo "osc6_ts rc=$?"; tail -c 600 gpurun_out/tp/bench_osc6_ts.json
