#!/bin/bash
# like ab.sh, but first checks each variant against the oracle on the small parity scenes: scripts/ab_check.sh <bench args> -- libs...
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
for lib in "$@"; do
  PBRHIP_LIB=$(realpath $lib) python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "render_matches or hooks" 2>&1 | tail -1
done
bash scripts/ab.sh "${args[@]}" -- "$@"
