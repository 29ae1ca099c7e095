#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM traffic counters for the default bench.py command.
# Counters are collected in their own passes (kernel-trace only), as gpurun requires.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps ${STEPS:-200} --warmup ${WARMUP:-20} --no-cpu-baseline --no-host-inclusive"
python bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
