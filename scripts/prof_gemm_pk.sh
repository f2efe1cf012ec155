#!/bin/bash
# rocprofv3 for the pre-packed 128-row GEMM: kernel trace + two SQ counter passes (each its own run, no trace domains mixed in).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$REPO/gpurun_out/prof_gemm_pk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gemm_pk_once.py <<PY
import sys
sys.path.insert(0, "$REPO")
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0")
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K, mode) in ((2048, 4096, 4096, 61), (512, 11008, 4096, 61), (512, 4096, 4096, 62), (512, 11008, 4096, 69)):
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    L.tce_w4a16_set_debug_mode(mode)
    for i in range(12):
        capi.check(capi.w4a16_forward(lins[i % 3].desc(x, out), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
PY
run() { name=$1; shift; timeout 200 rocprofv3 "$@" --output-format csv -d $OUT/$name -o pmc -- python /tmp/gemm_pk_once.py > $OUT/$name.log 2>&1; }
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python /tmp/gemm_pk_once.py > $OUT/kt.log 2>&1
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run pmc_sq3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
grep -A12 "gemm_pk_kernel\|kernel stats" $OUT/summary.txt | head -150
