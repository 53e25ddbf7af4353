#!/bin/bash
mkdir -p gpurun_out
{
export REPS=3
export SCHED_CONFIGS='[{},{"PBRHIP_SUSP_TURNS":"16"},{"PBRHIP_SUSP_TURNS":"32"},{"PBRHIP_SUSP_TURNS":"48"},{"PBRHIP_STREAMS":"1"},{"PBRHIP_STREAMS":"2"},{"PBRHIP_STREAMS":"3"},{"PBRHIP_TRACE_BLOCKS_SMALL":"0,0"},{"PBRHIP_TRACE_BLOCKS_SMALL":"4,64000000"},{"PBRHIP_TRACE_BLOCKS_SMALL":"3,16000000"},{"PBRHIP_TAIL_PATHS":"131072"},{"PBRHIP_TAIL_PATHS":"524288"},{"PBRHIP_PIPE_STOP":"1"},{"PBRHIP_PIPE_STOP":"4"},{}]'
timeout 1500 python scripts/sched_ab.py ggx 2>&1 | grep "^{\|^!!"
export SCHED_CONFIGS='[{}]'
for lib in build/rf24/libpbrhip.so build/rf40/libpbrhip.so build/wt1/libpbrhip.so build/wt3/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep "^{\|^1/8"
done
} > gpurun_out/r6_tune2.txt 2>&1
cat gpurun_out/r6_tune2.txt
