#!/bin/bash
# A/B of O-tree variants (k_trace launches of a C2 render, one path group): scripts/r5_q8_ab.sh <spp> <variant> ...
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export SPP=${1:-16}; shift
bash scripts/kt.sh "Q tree" PBRHIP_WIDE8=0
for v in "$@"; do bash scripts/kt.sh "O $v" PBRHIP_LIB=build/q8_$v/libpbrhip.so; done
bash scripts/kt.sh "Q tree again" PBRHIP_WIDE8=0
