#!/bin/bash
# A/B bench of library variants on the GPU box: scripts/ab.sh <bench args> -- lib1.so lib2.so ...
args=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done
shift
for lib in "$@"; do
  PBRHIP_LIB=$(realpath $lib) python bench.py --no-cpu-baseline --no-live-pmc "${args[@]}" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
k = {k: round(v, 1) for k, v in (r.get('kernel_ms_per_step') or {}).items() if v}
print('%-28s' % '$lib', 'Msamples/s %.1f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], k)
"
done
