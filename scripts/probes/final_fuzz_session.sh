# fuzzers on the final binary, logs -> gpurun_out/fz/
mkdir -p gpurun_out/fz
timeout 900 python scripts/fuzz_w8a8.py 2500 23 > gpurun_out/fz/fuzz_w8a8.log 2>&1; tail -2 gpurun_out/fz/fuzz_w8a8.log
timeout 1200 python scripts/fuzz_w4a16.py 300 61 > gpurun_out/fz/fuzz_w4a16.log 2>&1; tail -2 gpurun_out/fz/fuzz_w4a16.log
