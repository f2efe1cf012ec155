mkdir -p gpurun_out/h6
timeout 900 python -m pytest tests/test_gpu_w8a8.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/h6/test.log
SH="512x768x3072,512x768x768,108x768x3072,108x768x768,108x3072x768,16x768x3072,512x1024x4096,512x2048x8192,256x768x3072"
for v in A B A B; do TCE_LIB_PATH=$PWD/tinychatengine_amd/lib_ab/libtce_hip_$v.so W8A8_SHAPES=$SH timeout 300 python scripts/probes/w8a8_small_ab.py 2>/dev/null | tail -1 >> gpurun_out/h6/ab.jsonl; done
