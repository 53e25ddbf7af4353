"""VGPRs / scratch bytes of every kernel in a built library (reads the gfx950 code object): python scripts/kernel_table.py [lib.so] [prefix ...]"""
import os, re, subprocess, sys, tempfile


def table(lib):
    llvm = "/opt/rocm/lib/llvm/bin"
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run([llvm + "/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
    subprocess.run([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
    notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    tb = {}
    # one YAML map per kernel, keys in alphabetical order (.group_segment_fixed_size comes before .name): collect a block, file it at .symbol
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(name|symbol|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "symbol":
            if "name" in cur:
                tb[cur.pop("name")] = cur
            cur = {}
        elif m.group(1) == "name":
            cur["name"] = m.group(2)      # (argument names come first and are overwritten by the kernel's own)
        else:
            cur[m.group(1)] = int(m.group(2))
            if m.group(1) == "vgpr_count" and "name" in cur and cur["name"].startswith("_Z"):
                pass
    # .vgpr_count / .wavefront_size follow .symbol: second pass for them
    name = None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(name|vgpr_count):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            name = m.group(2)
        elif name in tb:
            tb[name]["vgpr_count"] = int(m.group(2))
    dem = subprocess.run(["c++filt"] + list(tb), check=True, capture_output=True, text=True).stdout.split("\n")
    return {d.replace("void pb::", "").replace("pb::", "").split("(")[0]: v for d, v in zip(dem, tb.values())}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbrlab_amd", "libpbrhip.so")
    pre = tuple(sys.argv[2:]) or ("k_",)
    for k, v in sorted(table(lib).items()):
        if k.startswith(pre):
            print(f"{k:42s} vgpr {v.get('vgpr_count'):4d}  scratch {v.get('private_segment_fixed_size'):4d}  lds {v.get('group_segment_fixed_size'):6d}")
