#!/usr/bin/env python3
"""VGPRs / scratch of the kernels of a built library: scripts/ktable.py [path/to/libpbrhip.so] [name prefix ...]"""
import os, re, subprocess, sys, tempfile, shutil
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbrlab_amd", "libpbrhip.so")
tmp = tempfile.mkdtemp()
try:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
finally:
    shutil.rmtree(tmp, ignore_errors=True)
table, cur, lds = {}, {}, 0
for line in notes.splitlines():   # (the keys of a kernel's record come in alphabetical order: .group_segment_fixed_size BEFORE .name)
    m = re.match(r"\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
    if m:
        if m.group(1) == "group_segment_fixed_size":
            lds = int(m.group(2))
        elif m.group(1) == "name":
            cur = table.setdefault(m.group(2), {})
            cur["group_segment_fixed_size"] = lds
        else:
            cur[m.group(1)] = int(m.group(2))
names = subprocess.run(["c++filt"] + list(table), check=True, capture_output=True, text=True).stdout.split("\n")
for d, v in sorted(zip(names, table.values())):
    n = d.replace("void pb::", "").replace("pb::", "").split("(")[0]
    if len(sys.argv) <= 2 or n.startswith(tuple(sys.argv[2:])):
        print(f"{n:50s} vgpr {v.get('vgpr_count'):4d} scratch {v.get('private_segment_fixed_size'):4d} lds {v.get('group_segment_fixed_size'):6d}")
