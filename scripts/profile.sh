#!/bin/bash
# rocprofv3 evidence for the dominant kernel (grouped gate+up GEMV launch): kernel-trace stats and, in SEPARATE passes,
# the PMC counters (never combined with tracing other than --kernel-trace, per the pool's rules).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r1}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --roofline-only --roofline-launches 128 --layers 8 --roofline-eager ${BENCH_EXTRA}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# keep only this library's rows of the big per-dispatch tables
for f in $(find $OUT -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
  (head -1 $f; grep "tce::" $f) > $f.tmp && mv $f.tmp $f
done
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
