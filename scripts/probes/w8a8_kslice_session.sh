# W8A8 forms beside the dispatcher's choice (weights rotating through HBM): W8A8_SHAPES=MxNxK,... and MODES="<debug mode> ..." ("" = the rule), e.g.
#   gpurun -- 'W8A8_SHAPES=512x768x3072,108x768x3072 MODES="0 19304 19404 19904 19001" bash scripts/probes/w8a8_kslice_session.sh'
# results -> gpurun_out/w8a8_forms/ab.jsonl (the round-6 sweeps behind the rules: profiles/r6/w8a8_kslice_ab.jsonl)
mkdir -p gpurun_out/w8a8_forms
SH=${W8A8_SHAPES:-512x768x3072,512x768x768,512x3072x768,108x768x3072,108x768x768,108x3072x768,16x768x3072}
for m in ${MODES:-0 19304 19404 19904 19001}; do
  [ "$m" = "0" ] && m=""
  W8A8_SHAPES=$SH W8A8_MODES=$m python scripts/probes/w8a8_small_ab.py 2>&1 | tail -1 >> gpurun_out/w8a8_forms/ab.jsonl
done
