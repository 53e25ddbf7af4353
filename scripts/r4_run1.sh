mkdir -p gpurun_out/r4a
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4a/tests.log
{
scripts/kt.sh base PBRHIP_TRACE2=0
scripts/kt.sh t2_5blk PBRHIP_TRACE2=1
scripts/kt.sh t2_4blk PBRHIP_TRACE2=1 PBRHIP_LIB=build/t2b4/libpbrhip.so
scripts/kt.sh base PBRHIP_TRACE2=0
scripts/kt.sh t2_5blk PBRHIP_TRACE2=1
scripts/kt.sh t2_4blk PBRHIP_TRACE2=1 PBRHIP_LIB=build/t2b4/libpbrhip.so
scripts/kt.sh t2_4blk_3 PBRHIP_TRACE2=1 PBRHIP_LIB=build/t2b4/libpbrhip.so PBRHIP_TRACE2_BLOCKS=3
python scripts/pvstats.py
} > gpurun_out/r4a/ab.log 2>&1
cat gpurun_out/r4a/tests.log gpurun_out/r4a/ab.log
