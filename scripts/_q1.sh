timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternative or tail or soup or stack" 2>&1 | tail -3
for v in ggx; do
for lib in pbrlab_amd/libpbrhip.so build/nooct/libpbrhip.so pbrlab_amd/libpbrhip.so build/nooct/libpbrhip.so; do
echo "== $lib $v"
PBRHIP_LIB=$(pwd)/$lib SCHED_CONFIGS='[{}]' REPS=5 python3 scripts/sched_ab.py $v 2>&1 | grep "world1\|per-kernel"
done; done
