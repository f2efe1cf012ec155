#!/bin/bash
# Round 3: one session, one box -- the bench line's per-launch-shape numbers (HIP events around graph replays) and, for the SAME command,
# rocprofv3 --kernel-trace --stats of the graph-replayed launches, then the PMC passes (separate runs, eager launches: counter collection
# serialises kernels anyway).  Never combines --pmc with any tracing but --kernel-trace (the pool's rule).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r3}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --shapes-only ${BENCH_EXTRA}"
timeout 300 $CMD > $OUT/shapes_graph.json 2> $OUT/shapes_graph.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD --roofline-eager > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD --roofline-eager > $OUT/pmc_sq.log 2>&1
for f in $(find $OUT -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
  (head -1 $f; grep "tce::" $f) > $f.tmp && mv $f.tmp $f
done
python $REPO/scripts/summarize_launch_shapes.py $OUT $OUT/shapes_graph.json > $OUT/summary_launch_shapes.txt 2>&1
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/summary_launch_shapes.txt
