mkdir -p gpurun_out/r4j
{
for i in 1 2; do
timeout 300 scripts/kt.sh pair6
timeout 300 scripts/kt.sh top0 PBRHIP_LIB=build/top0/libpbrhip.so
timeout 300 scripts/kt.sh top128 PBRHIP_LIB=build/top128/libpbrhip.so
timeout 300 scripts/kt.sh top256 PBRHIP_LIB=build/top256/libpbrhip.so
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|span" > gpurun_out/r4j/top.log
cat gpurun_out/r4j/top.log
