# 128 x 64 tiles (77: one quartet, 177: two) beside the dispatcher's choice where 128 x 64 tiles number 384 .. 511 (weights rotating through HBM); results -> gpurun_out/h13/ab.jsonl
mkdir -p gpurun_out/h13
SH="1024x3072x768,1024x3072x2048,1024x3072x3072,768x4096x1024,768x4096x4096,1024x3584x1024,896x3584x768,1024x3072x8192,640x5120x1280"
for m in "" 77 177 "" 77; do
  W8A8_SHAPES=$SH W8A8_MODES=$m python scripts/probes/w8a8_small_ab.py 2>&1 | tail -1 >> gpurun_out/h13/ab.jsonl
done
