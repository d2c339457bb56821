#!/bin/bash
# Runtime knobs that could change the kernel-boundary cost inside a replayed graph (about 2 us per launch, 258 launches per step).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python scripts/ab_bench.py "base::" "devkarg1:HIP_FORCE_DEV_KERNARG=1:" "devkarg0:HIP_FORCE_DEV_KERNARG=0:" \
  "pktcap1:DEBUG_CLR_GRAPH_PACKET_CAPTURE=1:" "pktcap0:DEBUG_CLR_GRAPH_PACKET_CAPTURE=0:" "optflush0:AMD_OPT_FLUSH=0:" \
  "optflush3:AMD_OPT_FLUSH=3:" "cpwait1:GPU_STREAMOPS_CP_WAIT=1:" "hdpwa0:DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0:" > $O/ab13.log 2>&1; echo "ab rc=$?"
cat $O/ab13.log
