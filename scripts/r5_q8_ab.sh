#!/bin/bash
# A/B of O-tree variants (k_trace launches of a 16-spp C2 render, one path group)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export SPP=16
bash scripts/kt.sh "Q tree" PBRHIP_WIDE8=0
bash scripts/kt.sh "O tree default"
for v in w11 w23 b5 lds16; do bash scripts/kt.sh "O $v" PBRHIP_LIB=build/q8_$v/libpbrhip.so; done
bash scripts/kt.sh "O cost 1.5,1" PBRHIP_Q8_COST=1.5,1
bash scripts/kt.sh "O cost 0.7,1" PBRHIP_Q8_COST=0.7,1
bash scripts/kt.sh "O cost 1,2" PBRHIP_Q8_COST=1,2
bash scripts/kt.sh "O cost 1,0.5" PBRHIP_Q8_COST=1,0.5
bash scripts/kt.sh "Q tree again" PBRHIP_WIDE8=0
