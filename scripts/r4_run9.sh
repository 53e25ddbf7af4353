mkdir -p gpurun_out/r4i
{
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do
timeout 300 scripts/kt.sh base PBRHIP_LIB=build/base/libpbrhip.so
timeout 300 scripts/kt.sh pair6
timeout 300 scripts/kt.sh pair5 PBRHIP_LIB=build/pair5/libpbrhip.so
done
PBRHIP_PV_STATS=1 SPP=8 timeout 300 python scripts/pvstats.py 2>&1 | grep -v "pv steps\|^walk" | head -8
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" > gpurun_out/r4i/pair.log
cat gpurun_out/r4i/pair.log
