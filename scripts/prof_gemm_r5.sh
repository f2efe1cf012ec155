#!/bin/bash
# Round 5: rocprofv3 for the prefill GEMM forms on the packed copy -- kernel trace + three SQ counter passes (each its own run, --pmc never combined with another
# trace domain).  Forms: 128-row tiles (61: one quartet, two workgroups per CU; 690+60: the dispatcher's 128-row choice), the 256-row wave tiles (66: one quartet;
# 68: two quartets alternating the k-blocks of one tile).  usage: prof_gemm_r5.sh [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r5}
OUT=$REPO/gpurun_out/prof_gemm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gemm_once.py <<PY
import sys
sys.path.insert(0, "$REPO")
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0")
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(1)
cases = [(2048, 4096, 4096, (61,)), (2048, 4096, 4096, (66,)), (2048, 4096, 4096, (68,)), (512, 4096, 4096, (690, 60)), (512, 11008, 4096, (690, 60)), (4096, 4096, 4096, (68,)), (4096, 4096, 4096, (690, 60))]
for (M, N, K, modes) in cases:
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    for m_ in modes: L.tce_w4a16_set_debug_mode(m_)
    for i in range(12):
        capi.check(capi.w4a16_forward(lins[i % 3].desc(x, out), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(691)
    del lins
PY
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -o pmc -- python /tmp/gemm_once.py > $OUT/$name.log 2>&1; }
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python /tmp/gemm_once.py > $OUT/kt.log 2>&1
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
run pmc_sq3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/summary.txt | cut -c1-260
