{
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 600 python scripts/frame_ab.py "PBRHIP_FIRST_DIRECT=0" "PBRHIP_FIRST_DIRECT=1" 2>&1 | grep spp
PBRHIP_LIB=build/base/libpbrhip.so timeout 600 python scripts/frame_ab.py "base=1" 2>&1 | grep spp
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl"
