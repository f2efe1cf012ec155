#!/bin/bash
# rocprofv3 counters for the prefill GEMM (M=512, 4096x4096): kernel trace + two SQ passes (each its own run).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out/prof_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gemm_once.py <<PY
import sys, ctypes as C
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/scripts")
import torch
from tinychatengine_amd import capi
from tune import ring, dev
L = capi.lib()
M, N, K = 512, 4096, 4096
sets = ring(N, K, 128, min_bytes=1e8)
x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
for v in ((4, 1), (4, 2), (204, 1), (204, 2), (0, 0)):  # older kernel tiles, LDS-DMA tiles (one quartet by default at these ids? no: the launcher picks the form from the tile count), the dispatcher
    capi.set_gemm_config(*v)
    for i in range(12):
        s = sets[i % len(sets)]
        d = capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr())
        capi.check(L.tce_w4a16_forward(C.byref(d), None))
    torch.cuda.synchronize()
PY
run() { name=$1; shift; timeout 200 rocprofv3 "$@" --output-format csv -d $OUT/$name -o pmc -- python /tmp/gemm_once.py > $OUT/$name.log 2>&1; }
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python /tmp/gemm_once.py > $OUT/kt.log 2>&1
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
# (TA_* / TCP_* / TCC_* passes abort in rocprofv3 on this image -- signal 6 after the timeout -- and are not run)
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/summary.txt | tail -100
