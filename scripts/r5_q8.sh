#!/bin/bash
# round 5: the O tree (8-wide): parity subset, then A/B against the Q tree.  usage: scripts/r5_q8.sh [variants: ggx sss hair c5]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out
PBRHIP_WIDE8=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "soup or alternative or trace_hooks or brute_force or axis_aligned or curves_only" 2>&1 | tail -5
: > $out/r5_q8_probe.txt
for v in ${@:-ggx hair}; do VARIANT=$v SPP=${SPP:-8} timeout 600 python scripts/q8_probe.py 2>&1 | grep -v "^pbrhip: free" | tee -a $out/r5_q8_probe.txt; done
