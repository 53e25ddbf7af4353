#!/bin/bash
# scripts/ab_sched.sh <variant> -- lib1.so lib2.so ...: schedule A/B (scripts/sched_ab.py) per library variant
variant=$1; shift; shift
for lib in "$@"; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) python scripts/sched_ab.py $variant 2>&1 | grep -v amdgpu.ids
done
