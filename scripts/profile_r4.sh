#!/bin/bash
# Round 4: one session, one box.  For BOTH shape sets the metric is read on (Llama-3-8B's true shapes = the headline; the set BASELINE.json spells):
#   * bench.py --shapes-only: the per-launch-shape numbers of the bench line (HIP events around graph replays),
#   * rocprofv3 --kernel-trace --stats of the SAME command (graph-replayed launches): per-kernel durations,
#   * the PMC passes (separate runs, eager launches: counter collection serialises kernels anyway): FETCH_SIZE; the SQ wave-cycle split.
# And for the dominant launch (grouped gate+up of the headline workload): the roofline command's kernel trace + FETCH_SIZE -> traffic.json, keyed to the
# kernel's sources (bench.py reports roofline.traffic only while they are unchanged).  Never combines --pmc with any tracing but --kernel-trace (the pool's rule).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
for WL in ${PROFILE_WORKLOADS-llama3-8b baseline-named}; do  # (PROFILE_WORKLOADS="" : the dominant launch only)
  OUT=$REPO/gpurun_out/prof_${TAG}_${WL}
  mkdir -p $OUT
  CMD="python $REPO/bench.py --shapes-only --workload $WL ${BENCH_EXTRA}"
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
done
# the dominant launch of the headline workload
OUT=$REPO/gpurun_out/prof_${TAG}_dominant
mkdir -p $OUT
CMD="python $REPO/bench.py --roofline-only --roofline-launches 128 --roofline-eager ${BENCH_EXTRA}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --roofline-only --roofline-launches 256 ${BENCH_EXTRA} > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
TCE_ALGO_BYTES=60628992 python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
for f in $(find $OUT -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
  (head -1 $f; grep "tce::" $f) > $f.tmp && mv $f.tmp $f
done
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
