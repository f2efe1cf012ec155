# two BUILDS of the library (tinychatengine_amd/lib_ab/libtce_hip_A.so / _B.so) on the prefill shapes, alternating (PK_PAIRS=1: the gate+up form with the SiLU * mul epilogue);
# the in-tree build's parity tests first
mkdir -p gpurun_out/h9
timeout 900 python -m pytest tests/test_gpu_w4a16_pk.py tests/test_gpu_epilogues.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/h9/test.log
for v in A B A B A B; do TCE_LIB_PATH=$PWD/tinychatengine_amd/lib_ab/libtce_hip_$v.so PK_SHAPES=${PK_SHAPES:-512x4096x4096,512x11008x4096,512x4096x11008,384x4096x4096,1024x4096x4096,512x4096x14336,256x4096x4096} timeout 300 python scripts/probes/gemm_pk_shapes.py 2>/dev/null | tail -1 >> gpurun_out/h9/ab.jsonl; done
