mkdir -p gpurun_out/r4b
PBRHIP_TRACEQ=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4b/tests.log
{
timeout 300 scripts/kt.sh base PBRHIP_TRACEQ=0
timeout 300 scripts/kt.sh pool PBRHIP_TRACEQ=1
timeout 300 scripts/kt.sh base PBRHIP_TRACEQ=0
timeout 300 scripts/kt.sh pool PBRHIP_TRACEQ=1
} > gpurun_out/r4b/ab.log 2>&1
cat gpurun_out/r4b/tests.log gpurun_out/r4b/ab.log
