mkdir -p gpurun_out/r4n
{
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|rel L2|c[1-5]:" | tail -30
timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1
for i in 1 2; do
timeout 300 scripts/kt.sh base PBRHIP_LIB=build/base/libpbrhip.so
timeout 300 scripts/kt.sh new
done
VARIANT=sss SPP=64 timeout 300 scripts/kt.sh base_c3 PBRHIP_LIB=build/base/libpbrhip.so
VARIANT=sss SPP=64 timeout 300 scripts/kt.sh new_c3
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" > gpurun_out/r4n/full.log
cat gpurun_out/r4n/full.log
