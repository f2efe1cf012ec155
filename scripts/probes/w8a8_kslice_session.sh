# w8a8_kslice_kernel: the W8A8 parity file, then the dispatcher's choice beside the forms forced off / on (weights rotating through HBM); results -> gpurun_out/h3/
mkdir -p gpurun_out/h3
python -m pytest tests/test_gpu_w8a8.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/h3/test.log
SH="512x768x3072,512x768x768,512x3072x768,108x768x3072,108x768x768,108x3072x768,16x768x3072,16x768x768,16x3072x768,512x2048x2048,512x2048x8192,512x1024x4096,1024x768x3072,2048x768x3072,512x1536x6144,108x2048x8192,16x2048x8192"
for m in "" 19001; do
  W8A8_SHAPES=$SH W8A8_MODES=$m python scripts/probes/w8a8_small_ab.py 2>&1 | tail -1 >> gpurun_out/h3/ab.jsonl
done
python scripts/fuzz_w8a8.py 400 > gpurun_out/h3/fuzz.log 2>&1; tail -3 gpurun_out/h3/fuzz.log
