# two BUILDS of the library (tinychatengine_amd/lib_ab/libtce_hip_A.so / _B.so) on W8A8 shapes, alternating (W8A8_OUT=fp32: the fp32-output form accumulating into C);
# the in-tree build's W8A8 parity file first
mkdir -p gpurun_out/h6
timeout 900 python -m pytest tests/test_gpu_w8a8.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/h6/test.log
SH=${W8A8_AB_SHAPES:-512x768x768,512x768x3072,512x3072x768,108x768x3072,2048x4096x4096,2048x4096x16384,512x4096x4096,2048x768x3072,512x4096x2048}
for v in A B A B; do TCE_LIB_PATH=$PWD/tinychatengine_amd/lib_ab/libtce_hip_$v.so W8A8_SHAPES=$SH timeout 300 python scripts/probes/w8a8_small_ab.py 2>/dev/null | tail -1 >> gpurun_out/h6/ab.jsonl; done
