mkdir -p gpurun_out/r4k
{
for i in 1 2; do
timeout 300 scripts/kt.sh new
for v in wt1 wt3 rf24 rf40; do timeout 300 scripts/kt.sh $v PBRHIP_LIB=build/$v/libpbrhip.so; done
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|span" > gpurun_out/r4k/knobs.log
cat gpurun_out/r4k/knobs.log
