#!/bin/bash
# round-5 diagnosis run (GPU box): micro-benchmark calibration + the same counters on an 8-spp C2 render + the k_trace launch list
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out
bash scripts/ubench/run_r5.sh > gpurun_out/ubench.log 2>&1
SPP=8 PASS_TIMEOUT=200 python3 scripts/pmc_adhoc.py c2 \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
  > $out/r5_c2_8spp_counters_before.txt 2>&1
SPP=8 bash scripts/kt.sh "c2 8spp baseline" > $out/r5_kt_baseline.txt 2>&1
SPP=64 bash scripts/kt.sh "c2 64spp baseline" >> $out/r5_kt_baseline.txt 2>&1
cat $out/r5_kt_baseline.txt
