mkdir -p gpurun_out/r4l
{
for i in 1 2; do
VARIANT=sss SPP=64 timeout 300 scripts/kt.sh base_c3 PBRHIP_LIB=build/base/libpbrhip.so
VARIANT=sss SPP=64 timeout 300 scripts/kt.sh new_c3
done
VARIANT=hair SPP=32 timeout 300 scripts/kt.sh base_c4 PBRHIP_LIB=build/base/libpbrhip.so
VARIANT=hair SPP=32 timeout 300 scripts/kt.sh new_c4
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" > gpurun_out/r4l/c3c4.log
cat gpurun_out/r4l/c3c4.log
