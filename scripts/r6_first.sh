#!/bin/bash
# round 6, first GPU check of the resumable rays + pipelined host loop: parity subset, then the schedule A/B (frame + eighth)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_first_parity.txt
cat gpurun_out/r6_first_parity.txt
export REPS=3
export SCHED_CONFIGS='[{"PBRHIP_SUSP_TURNS":"0","PBRHIP_PIPE_DEPTH":"1","PBRHIP_SHADOW_FIRST":"0"},
 {"PBRHIP_SUSP_TURNS":"0","PBRHIP_PIPE_DEPTH":"2","PBRHIP_SHADOW_FIRST":"0"},
 {"PBRHIP_SUSP_TURNS":"0","PBRHIP_PIPE_DEPTH":"2","PBRHIP_SHADOW_FIRST":"1"},
 {"PBRHIP_SUSP_TURNS":"16","PBRHIP_PIPE_DEPTH":"2"},
 {"PBRHIP_SUSP_TURNS":"32","PBRHIP_PIPE_DEPTH":"2"},
 {"PBRHIP_SUSP_TURNS":"64","PBRHIP_PIPE_DEPTH":"2"},
 {"PBRHIP_SUSP_TURNS":"128","PBRHIP_PIPE_DEPTH":"2"},
 {"PBRHIP_SUSP_TURNS":"32","PBRHIP_PIPE_DEPTH":"3"},
 {"PBRHIP_SUSP_TURNS":"32","PBRHIP_PIPE_DEPTH":"2","PBRHIP_PIPE_STOP":"0"},
 {"PBRHIP_SUSP_TURNS":"32","PBRHIP_PIPE_DEPTH":"2","PBRHIP_SHADOW_FIRST":"0"},
 {"PBRHIP_SUSP_TURNS":"32","PBRHIP_PIPE_DEPTH":"1"}]'
timeout 900 python scripts/sched_ab.py ggx > gpurun_out/r6_first_sched.txt 2>&1
tail -40 gpurun_out/r6_first_sched.txt
