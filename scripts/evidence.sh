#!/bin/bash
# Refresh of the judged evidence on a GPU box (writes gpurun_out/ev_<tag>/; copy what is to be judged into profiles/).
#   1. PMC passes for the HBM traffic of one full C2 render (FETCH_SIZE, WRITE_SIZE; counters only)  -> profiles/r1_c2_hbm_traffic_pmc.json
#   2. the driver's bench command, C2 with CPU baseline; C3 / C4 without                              -> r1_c{2,3,4}_bench.json
#   3. rocprofv3 --kernel-trace --stats of the same command (default: two path groups) and with --streams 1
#   4. SQ counter passes on an 8 spp render
# usage: scripts/evidence.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_$tag; mkdir -p $out
SPP=64 bash scripts/pmc_traffic.sh $tag > $out/traffic.log 2>&1
cp gpurun_out/traffic_$tag/summary.json $out/hbm_traffic_pmc.json && cp $out/hbm_traffic_pmc.json profiles/r1_c2_hbm_traffic_pmc.json
python3 bench.py --steps 3 --warmup 1 > $out/c2_bench.json 2> $out/c2_bench.err
python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $out/c3_bench.json 2> $out/c3_bench.err
python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $out/c4_bench.json 2> $out/c4_bench.err
for v in default solo; do
  extra=""; [ $v = solo ] && extra="--streams 1"
  rocprofv3 --kernel-trace --stats -f csv -d $out/prof_$v -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > $out/prof_$v.log 2>&1
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_$v.csv
  rm -rf $out/prof_$v
done
SPP=8 bash scripts/pmc2.sh $tag "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
   "SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" > $out/pmc_sq.log 2>&1
cp gpurun_out/pmc_$tag/summary.txt $out/pmc_sq_8spp_summary.txt
tail -1 $out/c2_bench.json | cut -c1-400
