# w8a8_kslice_kernel forced on the larger OPT widths beside the dispatcher's choice (weights rotating through HBM); results -> gpurun_out/h8/ab.jsonl
mkdir -p gpurun_out/h8
SH="512x4096x4096,512x16384x4096,512x4096x16384,512x8192x2048,512x2048x2048,2048x4096x4096,108x4096x4096,108x4096x16384,108x16384x4096,16x4096x4096,16x16384x4096"
for m in "" 19904 19404 19304; do
  W8A8_SHAPES=$SH W8A8_MODES=$m python scripts/probes/w8a8_small_ab.py 2>&1 | tail -1 >> gpurun_out/h8/ab.jsonl
done
