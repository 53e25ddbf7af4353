mkdir -p gpurun_out/r4m
{
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do
timeout 300 scripts/kt.sh base PBRHIP_LIB=build/base/libpbrhip.so
timeout 300 scripts/kt.sh new
done
VARIANT=sss SPP=64 timeout 300 scripts/kt.sh base_c3 PBRHIP_LIB=build/base/libpbrhip.so
VARIANT=sss SPP=64 timeout 300 scripts/kt.sh new_c3
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" > gpurun_out/r4m/glossy.log
cat gpurun_out/r4m/glossy.log
