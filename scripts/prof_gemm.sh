#!/bin/bash
# rocprofv3 counters for the prefill GEMM (M=512): where do the cycles go?
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
for v in ((4, 1), (4, 2)):
    capi.set_gemm_config(*v)
    for i in range(12):
        s = sets[i % len(sets)]
        d = capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr())
        capi.check(L.tce_w4a16_forward(C.byref(d), None))
    torch.cuda.synchronize()
PY
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_LDS[A-Z_0-9_]*\|SQ_INST_CYCLES[A-Z_0-9_]*\|SQ_WAIT[A-Z_0-9_]*\|SQ_ACTIVE[A-Z_0-9_]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/sq_names.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python /tmp/gemm_once.py > $OUT/kt.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python /tmp/gemm_once.py > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -o pmc -- python /tmp/gemm_once.py > $OUT/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_sq3 -o pmc -- python /tmp/gemm_once.py > $OUT/pmc_sq3.log 2>&1
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/sq_names.txt | head -c 3000; echo; cat $OUT/summary.txt | tail -70; tail -3 $OUT/pmc_sq2.log $OUT/pmc_sq3.log
