#!/bin/bash
# Round-5 calibration of the `k_trace` ceilings.  On the GPU box:
#   bash scripts/ubench/run_r5.sh            -> gpurun_out/profiles/r5_valu_rate.txt, r5_valu_rate2.txt, r5_gather2.txt, r5_ubench_pmc.txt
# 1. builds the micro-benchmarks (the binaries are git-ignored; nothing else builds them)
# 2. runs them plainly (events)
# 3. runs them under rocprofv3 --pmc, counters only, one pass per counter group (never next to a trace flag), the program itself after `--`
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out
for b in valu_rate valu_rate2 vmem_gather2 exec_halves vmem_quads; do
  [ -x scripts/ubench/$b ] && [ scripts/ubench/$b -nt scripts/ubench/$b.hip ] || hipcc --offload-arch=gfx950 -O3 scripts/ubench/$b.hip -o scripts/ubench/$b 2>&1 | grep -v hip-link
done
timeout 300 scripts/ubench/valu_rate > $out/r5_valu_rate.txt 2>&1; echo "valu_rate rc $?" >> $out/r5_valu_rate.txt
timeout 300 scripts/ubench/valu_rate2 > $out/r5_valu_rate2.txt 2>&1; echo "valu_rate2 rc $?" >> $out/r5_valu_rate2.txt
timeout 900 scripts/ubench/vmem_gather2 > $out/r5_gather2.txt 2>&1; echo "vmem_gather2 rc $?" >> $out/r5_gather2.txt
# counters, per kernel (summed over its dispatches): one rocprofv3 run per group
pmc() {  # pmc <label> <binary + args> -- <counters...>
  local label=$1 bin=$2 arg=$3; shift 3
  local d=/tmp/ub_pmc_$$; rm -rf $d
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" -f csv -d $d -o p -- $bin $arg > $d.out 2>&1) || { echo "$label [$*]: FAILED"; tail -3 $d.out; }
  python3 - "$d" "$label" "$*" <<'PY'
import csv, glob, sys, collections
# per kernel, its dispatches in order; the gather benchmark ("quick": tables 16 KB / 2 MB / 64 MB / 2 GB x 1 and 6 blocks per CU x 2 repetitions
# per kernel) is printed per (table, blocks per CU), everything else summed
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0]
        rows.setdefault(k, collections.OrderedDict()).setdefault(int(row["Dispatch_Id"]), collections.OrderedDict())[row["Counter_Name"]] = float(row["Counter_Value"])
tabs, bpcs = ["16 KB (L1)", "2 MB (L2)", "64 MB (MALL)", "2 GB (HBM)"], [1, 6]
for k, disp in rows.items():
    ids = sorted(disp)
    groups = [("", ids)]
    if sys.argv[2] == "gather" and len(ids) == 16:
        groups = [("%-13s %d blocks/CU" % (tabs[i // 4], bpcs[(i // 2) % 2]), ids[i:i + 2]) for i in range(0, 16, 2)]
    for label, g in groups:
        a = collections.OrderedDict()
        for i in g:
            for c, v in disp[i].items():
                a[c] = a.get(c, 0.0) + v
        print("%s | %-22s %-26s n=%-3d %s" % (sys.argv[2], k[:22], label, len(g), " ".join("%s=%.6g" % kv for kv in a.items())))
PY
  rm -rf $d $d.out
}
{
  R=$(pwd)
  pmc valu $R/scripts/ubench/valu_rate "" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc valu $R/scripts/ubench/valu_rate "" SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE
  pmc gather $R/scripts/ubench/vmem_gather2 quick TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  pmc gather $R/scripts/ubench/vmem_gather2 quick TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE   # (with the TA_ADDR_STALLED_BY_* counters next to them rocprofv3 aborts on this box)
  pmc gather $R/scripts/ubench/vmem_gather2 quick TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE
  pmc gather $R/scripts/ubench/vmem_gather2 quick SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
} > $out/r5_ubench_pmc.txt 2>&1
tail -n 40 $out/r5_valu_rate.txt
