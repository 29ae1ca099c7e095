#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats, HBM traffic counters and SQ counters for one bench.py workload.
#   WORKLOAD=c2|c3|c4|c5 (default c2)  STEPS / WARMUP (defaults per workload)
#   EXTRA="--mpa triple_speed" / "--seed 2" ... : further bench.py arguments (SURVEY.md 8(d)'s grid);  NAME=c2triple: what the summaries are
#   called (default: the workload; no underscores);  ROUND=r06;  SQ=0 skips the four SQ counter passes
# Every counter group is collected in a pass of its own (kernel-trace only), as gpurun requires; the program itself
# follows `--` (python3 bench.py ...), no wrapper in between.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${WORKLOAD:-c2}
case $W in
  c2) S=${STEPS:-200}; WU=${WARMUP:-20};;
  c3) S=${STEPS:-80}; WU=${WARMUP:-8};;
  c4) S=${STEPS:-24}; WU=${WARMUP:-8};;
  c5) S=${STEPS:-40}; WU=${WARMUP:-8};;
esac
NAME=${NAME:-$W}
OUT=gpurun_out/prof_$NAME
rm -rf $OUT && mkdir -p $OUT
ARGS="--workload $W --steps $S --warmup $WU --no-cpu-baseline --no-host-inclusive --no-scaling-reference ${EXTRA:-}"
echo "$ARGS" > $OUT/args.txt
python bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
if [ "${SQ:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- python3 bench.py $ARGS > $OUT/bench_sq1.json 2> $OUT/sq1.err
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq3 -- python3 bench.py $ARGS > $OUT/bench_sq3.json 2> $OUT/sq3.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq4 -- python3 bench.py $ARGS > $OUT/bench_sq4.json 2> $OUT/sq4.err
fi
# only the summaries travel back (the raw traces can be large): counter CSVs of the search / helper kernels, stats, trace
python tools/summarize_profiles.py ${ROUND:-r06}_$NAME $S $OUT > $OUT/summary_stdout.txt 2>&1
mkdir -p $OUT/profiles && cp profiles/${ROUND:-r06}_${NAME}_* profiles/${ROUND:-r06}_pmc_traffic_${NAME}.json $OUT/profiles/ 2>/dev/null  # (profiles/ of the box's copy does not travel back: gpurun_out/ does)
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT | tail -1
