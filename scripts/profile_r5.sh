#!/bin/bash
# Round 5: counter evidence for EVERY kernel family, one summary file each.  One driver process per pass (scripts/prof_families_once.py runs the
# prefill GEMM at M = 512 / 2048, the W8A8 GEMM on the OPT-125M shapes + 512 x 4096 x 4096, the fast attention step at 512 / 2048 keys and one whole decode token with
# its fused-norm launches); passes: rocprofv3 --kernel-trace --stats, then the PMC sets, each its OWN run and never beside any tracing but the kernel trace (the pool's rule).
# A case = (kernel name, grid size).  usage: profile_r5.sh [tag] [families...]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5}
shift
FAMS="${@:-gemm w8a8 attn token shard}"
OUT=$REPO/gpurun_out/prof_${TAG}_families
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROF_GEMM_FORMS=${PROF_GEMM_FORMS:-1,8}
CMD="python $REPO/scripts/prof_families_once.py $FAMS"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
run() { name=$1; shift; PROF_REPS=4 timeout 600 rocprofv3 "$@" --output-format csv -d $OUT/$name -o pmc -- $CMD > $OUT/$name.log 2>&1; }
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
for f in $(find $OUT -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
  (head -1 $f; grep "tce::" $f) > $f.tmp && mv $f.tmp $f
done
summ() { TCE_PROF_KEY_GRID=1 TCE_PROF_FILTER="$2" python $REPO/scripts/summarize_prof.py $OUT > $OUT/rocprofv3_summary_$1.txt 2>&1; }
summ gemm_pk "w4a16_gemm_pk"
summ w8a8 "w8a8_"
summ attention_step "attn_|attention"
summ decode_token_launches "w4a16_gemv|rmsnorm|glue|add_half|silu"
summ sharded_block_launches "w4a16_gemv_i8"
summ all ""
find $OUT -name "*.csv" -size +2M -delete
tail -5 $OUT/kt.log; wc -l $OUT/rocprofv3_summary_*.txt
