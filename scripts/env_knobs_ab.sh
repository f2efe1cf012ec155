#!/bin/bash
# Does any HIP runtime knob move the stream-ordered token (129 launches in one hipGraph)?  Same box, back to back, three runs each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { for i in 1 2 3; do env "$@" python bench.py --no-extras --no-cpu-baseline --issue graph --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'])"; done | tr '\n' ' '; }
echo "default: $(run X=1)"
echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1: $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1)"
echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)"
echo "AMD_OPT_FLUSH=0: $(run AMD_OPT_FLUSH=0)"
echo "AMD_OPT_FLUSH=1: $(run AMD_OPT_FLUSH=1)"
echo "ROC_SYSTEM_SCOPE_SIGNAL=0: $(run ROC_SYSTEM_SCOPE_SIGNAL=0)"
echo "AMD_DIRECT_DISPATCH=0: $(run AMD_DIRECT_DISPATCH=0)"
echo "DEBUG_HIP_GRAPH_BATCH_SIZE=256: $(run DEBUG_HIP_GRAPH_BATCH_SIZE=256)"
echo "ROC_ACTIVE_WAIT_TIMEOUT=0: $(run ROC_ACTIVE_WAIT_TIMEOUT=0)"
echo "HIP_FORCE_DEV_KERNARG=1: $(run HIP_FORCE_DEV_KERNARG=1)"
echo "HIP_FORCE_DEV_KERNARG=0: $(run HIP_FORCE_DEV_KERNARG=0)"
echo "GPU_MAX_HW_QUEUES=1: $(run GPU_MAX_HW_QUEUES=1)"
