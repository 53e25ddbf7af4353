#!/usr/bin/env python3
"""bench.py -- Msamples/s of the MI355X path-tracing hot path (BASELINE.json metric).

One "step" = one pbrlab::Render() of the workload: every (pixel, pass) sample of the frame goes
through the whole hot path (camera ray -> GetRadiance bounce loop -> RenderLayer).  The scene (BVH,
materials, light tables) is already resident in HBM when the timed region starts; the step ends with
the complete framebuffer in the HBM of rank 0.  pbrlab's RenderLayer is host memory: the same steps with the copy to the host
(PCIe-inclusive) are timed separately and reported as the secondary object "host_layer", never as `value`.

N GPUs: one process per GPU (launched by torch.distributed.run; torch.distributed is used for the rendezvous, the
barrier and the max-over-ranks of the time).  16x16 pixel blocks are dealt to the ranks (block i -> rank i % N), every
rank renders its blocks of the SAME 1920x1080 x 64 spp frame into a zeroed full-size device framebuffer ("scaling":
"strong": BASELINE's metric is this frame at 1/2/4/8 GPUs), and the library's own RCCL communicator (pbrhip_comm_*,
include/pbrhip.h) lands the frame on rank 0: by default a gather of the ranks' shards (every shard over its own xGMI link;
--exchange reduce = ncclReduce of the full layers, --exchange torch = torch.distributed.reduce).  The weak-scaling figure
(spp = 64 x N, per-GPU work fixed) is measured in the same run and reported as the secondary object "weak".

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1..4] on the procedural stand-in scenes (assets are missing: BASELINE.md §3)
    "c2": dict(desc="cornellbox_suzanne_lucy stand-in (S-cornell, reference .mtl, Lucy subsurface=0): "
                    "Lambert + GGX, 1920x1080, 64 spp", scene="cornell", variant="ggx", width=1920, height=1080, spp=64),
    "c3": dict(desc="S-cornell with Lucy subsurface=1 (random-walk SSS), 1920x1080, 256 spp", scene="cornell",
               variant="sss", width=1920, height=1080, spp=256),
    "c4": dict(desc="S-hair (50k curly strands) + head, PrincipledHair + curve intersection, 1920x1080, 128 spp",
               scene="hair", variant="", width=1920, height=1080, spp=128),
    "c1": dict(desc="S-cornell Lambert-only, 256x256, 4 spp (plumbing config)", scene="cornell", variant="lambert",
               width=256, height=256, spp=4),
    "c5": dict(desc="S-cornell (SSS) + S-hair, 3840x2160, 1024 spp (BASELINE configs[4]; 8.5 G samples: meant for 8 GPUs)",
               scene="cornell_hair", variant="sss", width=3840, height=2160, spp=1024),
}

# algorithmic bytes of the traversal kernel k_trace: 64 B per binary / Q-tree node visited (80 B per 8-wide O-tree node), 48 B per
# triangle tested, 64 B per curve tested; per closest-hit ray 32 B ray + 16 B hit record + 4 B queue entry; per
# shadow ray 32 B ray (origin shared with the continuation ray) + 4 B queue entry + 16 B pending contribution
NODE_B, TRI_B, CURVE_B, RAY_B, SHADOW_RAY_B = 64, 48, 64, 52, 52
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


SIMDS, XCDS, CUS, CLOCK_HZ = 1024, 8, 256, 2.4e9   # 256 CUs x 4 SIMDs in 8 XCDs; MI355X_MICROARCH.md peak engine clock (the measured clock is reported next to it)
# Per-lane 16-byte gathers through the vector L1 (every lane its own 64-byte item, four global_load_dwordx4; the table resident in
# L1): 23.9 bytes per clock and CU, measured with one 16 KB table for all blocks at 1 / 2 / 4 / 6 blocks per CU
# (scripts/ubench/vmem_gather2.hip, profiles/r5_gather2.txt; L2-resident: 21.9; a fully coalesced control: 31.7).  An ESTIMATE of
# what the load path can deliver to k_trace's access pattern -- it never sets `bound`.
GATHER_L1_BYTES_PER_CLK_CU = 23.9
# ... by item footprint (the same file: own 64 B = 4 loads 23.9, own 80 B = 5 loads 28.0, own 128 B = 8 loads 16.0 bytes per clock and CU from L1)
GATHER_PEAK_BY_ITEM = {64: 23.9, 80: 28.0, 128: 16.0}


# Cycles one wave64 VALU instruction occupies its SIMD, by class -- MEASURED (scripts/ubench/valu_rate.hip, valu_rate2.hip: profiles/r5_valu_rate*.txt,
# r5_ubench_pmc.txt), in the chip's own cycles (GRBM_GUI_ACTIVE / 8 XCDs over SQ_INSTS_VALU / 1024 SIMDs of the VALU-bound calibration kernels:
# v_fma_f32 2.33, v_pk_fma 4.18, v_pk_mul 4.19, v_pk_add 4.26, v_min 4.20, v_cvt_f32_ubyte 4.32, v_mul_lo 4.23, v_rcp 8.18; the second table's
# instructions at the same 0.91 x nominal clock).  Round 6: the VALU ceiling is  SQ_INSTS_VALU x (the kernel's own mean cost per instruction,
# from the histogram of its main loop in the code object) / (1024 SIMDs x the kernel's cycles)  -- not SQ_ACTIVE_INST_VALU x 4, which prices every
# instruction at four cycles and read 1.16-1.25 on kernels made of plain fp32 multiply-adds.
VALU_FULL, VALU_HALF, VALU_TRANS, VALU_F64 = 2.33, 4.2, 8.2, 8.2
VALU_FULL_RATE = ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32",
                  "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_cndmask_b32", "v_add_co_u32", "v_addc_co_u32", "v_not_b32", "v_accvgpr")
VALU_TRANSCENDENTAL = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def valu_cost(mnemonic):
    m = mnemonic.lower()
    if m.endswith("_f64") or "_f64_" in m or m.startswith(("v_mul_hi_", "v_mad_u64", "v_mad_i64", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64")):
        return VALU_F64 if "f64" in m else VALU_HALF
    if m.startswith(VALU_TRANSCENDENTAL):
        return VALU_TRANS
    base = m[:-4] if m.endswith(("_e32", "_e64")) else m
    base = base.replace("_dpp", "").replace("_sdwa", "")
    if base.startswith(VALU_FULL_RATE):
        return VALU_FULL
    return VALU_HALF     # packed fp32, min / max / min3 / max3, compares, conversions, shifts, bit-field, integer multiply / mad, readlane ...: 3.9-4.5 measured


def isa_histogram(kernel_substring):
    """Static instruction histogram of one kernel of the built libpbrhip.so (llvm-objdump on its gfx950 code object): the VALU instructions of
    its MAIN LOOP -- the widest backward branch: the persistent traversal's loop -- by mnemonic, and their mean measured cost (valu_cost).
    A static mix stands in for the dynamic one (the loop's phases run at different frequencies): an estimate, said so in the output."""
    import collections
    import re
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.environ.get("PBRHIP_LIB") or os.path.join(ROOT, "pbrlab_amd", "libpbrhip.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(llvm, "llvm-objdump")) and shutil.which("c++filt")):
        return None
    tmp = tempfile.mkdtemp(prefix="pbr_isa_", dir="/tmp")
    try:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True, capture_output=True)
        asm = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = re.findall(r"^[0-9a-f]+ <(\S+)>:$", asm, re.M)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    want = next((n for n, d in zip(names, dem) if kernel_substring in d), None)
    if want is None:
        return None
    body = asm.split(f"<{want}>:\n", 1)[1]
    body = re.split(r"\n[0-9a-f]+ <\S+>:\n", body, 1)[0]
    rows = []   # (address, mnemonic, branch target or None)
    for line in body.splitlines():
        m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
        rows.append((int(m.group(2), 16), m.group(1), t.group(1) if t else None))
    if not rows:
        return None
    base = rows[0][0]
    lo, hi = rows[0][0], rows[-1][0]
    span = 0
    for addr, mn, tgt in rows:
        if tgt is not None and mn.startswith(("s_cbranch", "s_branch")):
            ta = base + int(tgt, 16)
            if ta < addr and addr - ta > span:
                span, lo, hi = addr - ta, ta, addr
    hist = collections.Counter(mn for addr, mn, _ in rows if lo <= addr <= hi and mn.startswith("v_"))
    n = sum(hist.values())
    if not n:
        return None
    cost = sum(valu_cost(mn) * c for mn, c in hist.items()) / n
    full = sum(c for mn, c in hist.items() if valu_cost(mn) == VALU_FULL)
    return {"kernel": next(d for nme, d in zip(names, dem) if nme == want).split("(")[0].replace("void pb::", ""), "valu_in_loop": n,
            "mean_cycles_per_valu": cost, "full_rate_share": full / n, "salu_in_loop": sum(1 for addr, mn, _ in rows if lo <= addr <= hi and mn.startswith("s_")),
            "vmem_in_loop": sum(1 for addr, mn, _ in rows if lo <= addr <= hi and mn.startswith(("global_", "flat_", "buffer_", "scratch_"))),
            "top": dict(hist.most_common(8))}


def csrc_hash():
    """identifies the kernel sources a profile was taken on: sha256 over the files of pbrlab_amd/csrc (names + contents;
    there is no .git on the GPU box).  scripts/profile_round.py stores it with the counters it collects."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "pbrlab_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".h", ".hip", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(base, name), "rb").read())
    for name in ("pbr_f64r.h", "pbr_glibcf.h", "pbrhip.h"):  # headers outside csrc the kernels are compiled from
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "include", name), "rb").read())
    return h.hexdigest()[:16]


def pmc_record(workload, spp, world, kernel):
    """Counters cannot be read live, so they come from the committed PMC passes of this round (profiles/r6_<workload>_pmc.json,
    written by scripts/profile_round.py: FETCH_SIZE, WRITE_SIZE and SQ counters in separate --pmc passes over one render of
    the same workload, per kernel).  They are only used when that file was collected on the very kernel sources that are
    running (csrc_hash) and on the same configuration; otherwise the counter-based fields are null."""
    path = os.path.join(ROOT, "profiles", f"r6_{workload}_pmc.json")
    if world != 1 or not os.path.exists(path):
        return None, "no PMC record for this configuration"
    rec = json.load(open(path))
    if rec.get("csrc_hash") != csrc_hash():
        return None, f"stale PMC record (collected on csrc {rec.get('csrc_hash')}, running {csrc_hash()})"
    if rec.get("spp") != spp:
        return None, "PMC record is for another spp"
    k = next((v for name, v in rec["kernels"].items() if name.startswith(kernel)), None)
    return k, rec.get("source", "")


# (one rocprofv3 run per group; FETCH_SIZE / WRITE_SIZE apart as the guide prescribes; 8 SQ slots, GRBM and TCP are their own blocks)
PMC_PASSES = ["FETCH_SIZE", "WRITE_SIZE",
              "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE",
              "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"]


def pmc_live(workload, spp, kernel, budget_s):
    """The counters of `kernel` measured by THIS run: rocprofv3 --pmc passes (counters only: no trace flag next to --pmc, one
    pass per counter group as MI355X_MICROARCH.md prescribes; FETCH_SIZE / WRITE_SIZE apart) over one render of the workload as
    one path group in child processes (scripts/render_once.py: the program itself after `--`), after the timed region.
    Returns (record like profiles/r6_<workload>_pmc.json's kernel entry, note) or (None, why) -- the caller then falls back
    to the committed record."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    variant = {"c2": "ggx", "c3": "sss", "c4": "hair"}.get(workload)
    if variant is None or not os.path.exists(exe):
        return None, "no live PMC passes (rocprofv3 not found or no single-scene workload)"
    t0 = time.time()
    env = dict(os.environ, TMPDIR="/tmp", VARIANT=variant, SPP=str(spp), PBRHIP_STREAMS="1", REPS="1")
    agg, disp = collections.defaultdict(float), set()
    tmp = tempfile.mkdtemp(prefix="pbr_pmc_", dir="/tmp")
    try:
        for i, counters in enumerate(PMC_PASSES):
            left = budget_s - (time.time() - t0)
            if left < 20:
                return None, f"live PMC passes over budget ({budget_s} s)"
            d = os.path.join(tmp, f"p{i}")
            cmd = [exe, "--pmc"] + counters.split() + ["-f", "csv", "-d", d, "-o", f"p{i}", "--", sys.executable, os.path.join(ROOT, "scripts", "render_once.py")]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"live PMC pass {counters.split()[0]} timed out"
            if r.returncode != 0:
                return None, f"live PMC pass {counters.split()[0]} failed (rc {r.returncode})"
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if not row["Kernel_Name"].split("(")[0].replace("void ", "").startswith(kernel):
                        continue
                    agg[row["Counter_Name"]] += float(row["Counter_Value"])
                    if i == 0:
                        disp.add(row["Dispatch_Id"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not disp or "FETCH_SIZE" not in agg or "SQ_INSTS_VALU" not in agg:
        return None, "live PMC passes returned no rows for the kernel"
    rec = dict(agg)
    rec["dispatches"] = len(disp)
    rec["hbm_bytes_per_dispatch_fetch_x2"] = (2 * agg["FETCH_SIZE"] + agg.get("WRITE_SIZE", 0.0)) * 1024 / len(disp)
    return rec, (f"measured by this run: rocprofv3 --pmc passes {[p.split()[0] for p in PMC_PASSES]} over one {spp}-spp render of the workload "
                 f"(one path group, child processes, after the timed region, {time.time() - t0:.0f} s); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 on gfx950")


def make_desc(w):
    from pbrlab_amd import scenes
    if w["scene"] == "cornell":
        return scenes.cornell_scene(w["variant"], seed=1)
    if w["scene"] == "cornell_hair":
        return scenes.cornell_hair_scene(w["variant"], seed=1)
    return scenes.hair_scene(seed=1)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_run(workload, spp_override, seconds_budget):
    """Oracle (reference algorithm restated in C, oracle/) on the host cores, bounded sample of the same workload, with
    the reference's own worker pool: job = (tile, pass), job id -> (id % tiles, id / tiles) from one atomic counter
    (render.cc:210-233; oracle/pbr_oracle.c ORC_JOBS_TILE_PASS).  Three timings: one thread on every 6th tile at 1 spp
    (per-thread speed), all threads on 1/16 of the tiles (calibration), all threads on the full frame at as many spp as
    fit ~seconds_budget.  Which build of the oracle is loaded is decided by PBR_ORACLE_SO (tests/_oracle.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    w = dict(WORKLOADS[workload])
    desc = make_desc(w)
    so = O.oracle_scene_from_desc(desc)
    # the CPUs this job may really use: the GPU boxes report 256 hardware threads and run it under a cgroup quota of 16 CPUs
    # (cpu.max); threads = 2 x that quota -- the oracle peaks there (scripts/cpu_scaling.py: 1 / 16 / 32 / 256 threads = 0.34 / 4.9 /
    # 5.6 / 3.4 Msamples/s) -- and `cores` reports the quota
    quota = O.host_threads()
    cores = O.oracle_threads()
    W, H = w["width"], w["height"]
    t0 = time.time()                               # one thread: every 6th tile (a representative ~1/6 of the frame is too
    one_world = max(6, int(round(W * H / 400000)))  # long for big frames: about 0.4 M samples)
    _, _, st1 = so.render(W, H, 1, tile_rank=0, tile_world=one_world, threads=1, job_mode=O.JOBS_TILE_PASS)
    dt1 = max(time.time() - t0, 1e-3)
    rate1 = st1["samples"] / dt1
    t0 = time.time()                               # calibrate the pool on 1/16 of the tiles
    _, _, st = so.render(W, H, 1, tile_rank=0, tile_world=16, threads=cores, job_mode=O.JOBS_TILE_PASS)
    rate = st["samples"] / max(time.time() - t0, 1e-3)
    spp, world = 1, 1
    if rate * seconds_budget >= W * H:
        spp = int(max(1, min(w["spp"], rate * seconds_budget // (W * H))))
    else:
        world = int(min(64, max(1, round(W * H / (rate * seconds_budget)))))
    t0 = time.time()
    _, _, st = so.render(W, H, spp, tile_rank=0, tile_world=world, threads=cores, job_mode=O.JOBS_TILE_PASS)
    dt = time.time() - t0
    busy = float(so.L.orc_last_render_busy())
    valn = st["samples"] / dt / 1e6
    return {"value": valn, "unit": "Msamples/s", "cores": quota, "threads": cores, "hardware_threads": os.cpu_count(),
            "one_thread": {"value": rate1 / 1e6, "sample": f"{W}x{H} x 1 spp, tiles i%{one_world}==0: {st1['samples']} samples in {dt1:.1f} s on 1 thread"},
            "speedup_over_one_thread": valn / (rate1 / 1e6),
            "parallel_efficiency": valn / (rate1 / 1e6) / quota,
            "schedule_efficiency": busy / (cores * dt),
            "sample": f"{W}x{H} x {spp} spp, tiles i%{world}==0, jobs = (tile, pass) as render.cc:210-219: {st['samples']} samples in {dt:.1f} s on {cores} threads"}


def cpu_baseline(workload):
    """two builds of the oracle, each timed in its own process: the -O2 baseline-x86-64 build the parity tests use, and
    -O3 -march=native built here for this host; kind "port": Embree is not available, so the
    reference itself cannot run"""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    native = os.path.join(odir, "libpbr_oracle_native.so")
    builds = [("-O2 -ffp-contract=off (baseline x86-64: the checker's build)", os.path.join(odir, "libpbr_oracle.so"), 12.0)]
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-shared",
                               "-o", native, os.path.join(odir, "pbr_oracle.c"), "-lm", "-lpthread"])
        builds.append(("-O3 -march=native -ffp-contract=off (built on this host)", native, 10.0))
    except Exception as e:  # no compiler on the box: report the one build
        print(f"bench: native oracle build failed ({e})", file=sys.stderr)
    runs = []
    for flags, so_path, budget in builds:
        env = dict(os.environ, PBR_ORACLE_SO=so_path)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workload, "--cpu-seconds", str(budget)],
                             env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(out.stderr[-2000:], file=sys.stderr)
            continue
        r = json.loads(out.stdout.strip().splitlines()[-1])
        r["build"] = flags
        runs.append(r)
    if not runs:
        return None
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "Msamples/s", "cores": best["cores"], "threads": best["threads"],
            "hardware_threads": best["hardware_threads"], "kind": "port", "cpu": cpu_model(),
            "one_thread": best["one_thread"], "speedup_over_one_thread": best["speedup_over_one_thread"],
            "parallel_efficiency": best["parallel_efficiency"], "schedule_efficiency": best["schedule_efficiency"],
            "sample": "oracle = a SCALAR C restatement of the reference integrator over its own binned-SAH BVH2 (NOT Embree: Embree 4 is "
                      "not available here, so the reference itself cannot run; Embree's SSE/AVX BVH4 kernels are typically several "
                      "times faster per ray than a scalar BVH2, so the GPU / CPU ratio is over this port, not over the reference); "
                      "worker pool as render.cc:203-238 with the reference's job = (tile, pass); " + best["sample"] + "; build: " + best["build"]
                      + "; cores = the CPUs the job may use (min of hardware threads, affinity and the cgroup's cpu.max quota: the GPU boxes "
                      "have 256 hardware threads and give the job 16 CPUs), threads = what was started; parallel_efficiency = value / "
                      "(one-thread value x cores); schedule_efficiency = sum of the workers' busy wall time / (threads x wall time)",
            "builds": [{"build": r["build"], "value": r["value"], "one_thread": r["one_thread"]["value"],
                        "schedule_efficiency": r["schedule_efficiency"], "sample": r["sample"]} for r in runs]}


def spawn_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`
    as a child process (one fresh process per GPU, each reads RANK / LOCAL_RANK / WORLD_SIZE), pass its output through -- rank 0
    prints the JSON line -- and return its exit code.  Nothing in this process has imported torch or touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("BENCH_SPAWN_DRY_RUN"):   # (CPU test: what would be started)
        print(json.dumps({"spawn": cmd}))
        return 0
    return subprocess.run(cmd, env=env).returncode


def rank_diagnostics(dist, device, render_ms, exchange_ms):
    """N > 1: what each rank spent per step -- gathered AFTER the timed region -- so that a disappointing scaling figure can be read:
    per_rank_render_ms[r] = rank r's own Render() (its shard of the frame, blocking), exchange_ms = the slowest rank's time from the end
    of its render to the frame being complete on rank 0 (the wait for the slowest rank's render is part of it: see exchange_ms_rank0 for
    the root's own, i.e. the transfer), imbalance = max / mean of the render times.  Every rank calls it (one all_gather)."""
    import torch
    mine = torch.tensor([float(render_ms), float(exchange_ms)], dtype=torch.float64, device=device)
    allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, mine)
    render = [float(t[0]) for t in allr]
    exch = [float(t[1]) for t in allr]
    mean = sum(render) / len(render)
    return {"per_rank_render_ms": render, "exchange_ms": max(exch), "exchange_ms_per_rank": exch, "exchange_ms_rank0": exch[0],
            "imbalance": max(render) / mean if mean > 0 else None,
            "slowest_rank": max(range(len(render)), key=render.__getitem__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override spp (invalidates the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)   # internal: one oracle build, one JSON line
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic / valu from the committed PMC record instead of measuring them in this run")
    ap.add_argument("--pmc-budget", type=float, default=150.0, help="seconds the live rocprofv3 --pmc passes may take in all")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="concurrent path groups (0 = the library's default)")
    ap.add_argument("--shard-block", type=int, default=16,
                    help="edge of the pixel blocks dealt to the ranks at N > 1 (the reference's tile is 64; smaller balances better)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = the fixed 64-spp frame split over N GPUs (default: BASELINE's metric); "
                         "weak = spp x N (per-GPU work fixed).  The other one is reported as a secondary object")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce", "torch"],
                    help="N > 1: how the RenderLayer reaches rank 0: gather / reduce = RCCL inside libpbrhip "
                         "(pbrhip_comm_gather_layer / pbrhip_comm_reduce_layer), torch = torch.distributed.reduce")
    ap.add_argument("--spawn-selftest", action="store_true", help=argparse.SUPPRESS)   # CPU test of the launcher path (tests/test_dist_cpu.py)
    args = ap.parse_args()
    if args.spawn_selftest and "WORLD_SIZE" in os.environ:
        # a rank started by spawn_ranks(): rendezvous over gloo on 127.0.0.1, one collective, rank 0 prints the line (no GPU involved)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([dist.get_rank() + 1])
        dist.all_reduce(t)
        diag = rank_diagnostics(dist, torch.device("cpu"), 10.0 * (dist.get_rank() + 1), 1.0 + dist.get_rank())   # (the N > 1 line's per-rank fields)
        if dist.get_rank() == 0:
            print(json.dumps({"selftest": dist.get_world_size(), "sum": int(t.item()), "diagnostics": diag}))
        dist.destroy_process_group()
        return
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_run(args.cpu_baseline_child, 0, args.cpu_seconds)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N rank processes ourselves -- torch.distributed.run as a CHILD,
        # decided here, BEFORE this process imports torch or the library (a process that initialised the GPU must never fork a
        # launcher or be replaced by another program) -- hand on rank 0's JSON line and the launcher's exit code.
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import pbrlab_amd as pa
    from pbrlab_amd import api
    from pbrlab_amd.dist import reduce_layer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it with --nproc-per-node {args.gpus} (or without a launcher)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    pa.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w = dict(WORKLOADS[args.workload])
    if args.spp:
        w["spp"] = args.spp
    W, H = w["width"], w["height"]
    weak = args.scaling == "weak"
    spp = w["spp"] * (world if weak else 1)   # strong (default): the frame is fixed; weak: W*H/N pixels x 64*N spp per GPU
    desc = make_desc(w)
    scene = pa.scene_from_desc(desc)          # upload + BVH: outside the timed region
    info = scene.info()

    rgba = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    count = torch.zeros((H, W), dtype=torch.int32, device=dev)
    h_rgba = torch.zeros((H, W, 4), dtype=torch.float32).pin_memory() if rank == 0 else None
    h_count = torch.zeros((H, W), dtype=torch.int32).pin_memory() if rank == 0 else None
    torch.cuda.synchronize()
    ptrs = (rgba.data_ptr(), count.data_ptr())

    shard_block = args.shard_block if world > 1 else 0   # one rank: the library's default order (64 x 64 tiles)

    # the exchange step: the library's RCCL communicator (one rank makes the id, torch.distributed hands it round)
    comm, exchange = None, args.exchange
    if dist is not None and exchange != "torch":
        try:                                 # every rank checks that the library finds its RCCL before any collective call
            my_id, usable = pa.Comm.unique_id(), 1
        except Exception as e:
            print(f"bench: library communicator unavailable on rank {rank} ({e})", file=sys.stderr)
            my_id, usable = None, 0
        flag = torch.tensor([usable], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            uid = [my_id if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            try:
                comm = pa.Comm(uid[0], rank, world)  # ncclCommInitRank: collective
                ok = 1
            except Exception as e:                   # (an error every rank sees, e.g. an RCCL version mismatch: fall back together)
                print(f"bench: library communicator failed on rank {rank} ({e})", file=sys.stderr)
                comm, ok = None, 0
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag.item()):
                if comm is not None:
                    comm.close()                     # (created here, failed elsewhere: release the communicator and its stream)
                comm, exchange = None, "torch"
        else:
            exchange = "torch"                       # reported in config.exchange

    clock = {"render": 0.0, "exchange": 0.0}   # this rank's host clock per phase, summed over the steps of a timed() call

    def step(spp, flags=0, to_host=False):
        t_a = time.perf_counter()
        _, st = api.Render(scene, W, H, spp, tile_rank=rank, tile_world=world, device_out=ptrs, flags=flags,
                           max_paths_in_flight=args.max_paths, num_streams=args.streams, shard_block=shard_block)   # (blocking: the shard is complete in this rank's HBM)
        t_b = time.perf_counter()
        clock["render"] += t_b - t_a
        if dist is not None:                   # the only exchange step: the framebuffer, over xGMI
            if exchange == "gather":
                comm.gather_layer(scene, W, H, ptrs[0], ptrs[1], shard_block=shard_block, root=0)
            elif exchange == "reduce":
                comm.reduce_layer(ptrs[0], ptrs[1], W * H, root=0)
            else:
                reduce_layer(rgba, count, dst=0)
        if to_host and rank == 0:              # pbrlab's RenderLayer is host memory: the PCIe-inclusive figure ("host_layer"), never `value`
            h_rgba.copy_(rgba, non_blocking=True)
            h_count.copy_(count, non_blocking=True)
        torch.cuda.synchronize()
        clock["exchange"] += time.perf_counter() - t_b
        return st

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(spp, steps, warmup, flags, to_host=False):
        for _ in range(warmup):
            step(spp, to_host=to_host)
        barrier()
        clock["render"] = clock["exchange"] = 0.0
        t0 = time.perf_counter()
        agg = {}
        for _ in range(steps):
            st = step(spp, flags=flags, to_host=to_host)
            for k, v in st.items():
                agg[k] = agg.get(k, 0) + v
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, agg

    def check_layer(spp):   # sanity: every pixel of the frame on rank 0 got spp samples
        if rank == 0:
            h_rgba.copy_(rgba)
            h_count.copy_(count)
            torch.cuda.synchronize()
            assert int(h_count.min()) == spp and int(h_count.max()) == spp, "framebuffer incomplete"
            assert bool(torch.isfinite(h_rgba).all())

    # `value`: the frame is complete in rank 0's HBM when a step ends (the scene was resident when it started); the copy of the
    # RenderLayer to the host -- pbrlab's layer is host memory -- is measured separately below ("host_layer", PCIe-inclusive)
    # (HIP events around the k_trace launches only -- the roofline's kernel --: events around every launch cost the frame ~1 %)
    elapsed, agg = timed(spp, args.steps, args.warmup, 0 if args.no_roofline else api.RENDER_TIMING_TRACE)
    diag = rank_diagnostics(dist, dev, clock["render"] / args.steps * 1e3, clock["exchange"] / args.steps * 1e3) if dist is not None else None
    check_layer(spp)
    h_elapsed, _ = timed(spp, args.steps, 1, 0, to_host=True)   # (one untimed step first: the pinned destination is touched, the copy engine is warm)
    host_layer = {"value": W * H * spp * args.steps / h_elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": h_elapsed / args.steps * 1e3,
                  "note": "the same steps, each ending with the RenderLayer (rgba f32 + count u32: 20 bytes per pixel) copied to pinned host memory of rank 0 -- "
                          "pbrlab's RenderLayer is host memory; PCIe-inclusive, reported beside `value`, never as it"}
    other = None
    if world > 1:   # the other scaling mode, same run, as a secondary figure
        o_spp = w["spp"] * (1 if weak else world)
        o_elapsed, _ = timed(o_spp, args.steps, 1, 0)
        check_layer(o_spp)
        other = {"scaling": "strong" if weak else "weak", "spp": o_spp, "ms_per_step": o_elapsed / args.steps * 1e3,
                 "value": W * H * o_spp * args.steps / o_elapsed / 1e6, "unit": "Msamples/s"}

    roofline = None
    if not args.no_roofline:
        # untimed pass with traversal counters (deterministic: identical work to the timed steps)
        _, sst = api.Render(scene, W, H, spp, tile_rank=rank, tile_world=world, device_out=ptrs,
                            flags=api.RENDER_STATS, max_paths_in_flight=args.max_paths, num_streams=args.streams, shard_block=shard_block)
        torch.cuda.synchronize()
        # untimed: the same frame with ONE path group, i.e. every k_trace launch has the GPU to itself
        _, solo = api.Render(scene, W, H, spp, tile_rank=rank, tile_world=world, device_out=ptrs,
                             flags=api.RENDER_TIMING, max_paths_in_flight=args.max_paths, num_streams=1, shard_block=shard_block)
        torch.cuda.synchronize()
        node_b = int(sst.get("node_bytes") or NODE_B)    # 64: a node of the Q tree (4 children, quantised boxes) or of the binary tree
        curve_b = int(sst.get("curve_bytes") or CURVE_B)  # 32 when a curve piece is two 16-byte points of a chain (Q tree), 64 for a slot
        bytes_step = (node_b * (sst["closest_nodes"] + sst["shadow_nodes"]) + TRI_B * (sst["closest_tris"] + sst["shadow_tris"]) +
                      curve_b * (sst["closest_curves"] + sst["shadow_curves"]) + RAY_B * sst["closest_rays"] +
                      SHADOW_RAY_B * sst["shadow_rays"])
        launches = agg["n_trace_closest"] / args.steps
        ms_step = agg["ms_trace_closest"] / args.steps
        if ms_step > 0:
            achieved = bytes_step / (ms_step * 1e-3) / 1e9
            solo_ms = solo["ms_trace_closest"]
            solo_launch_s = solo_ms * 1e-3 / max(solo["n_trace_closest"], 1)
            # counter-based ceilings, from this round's committed PMC passes on these very kernel sources (else null)
            default_cfg = not (args.max_paths or args.streams or args.spp)
            pmc, pmc_note = (None, "non-default configuration")
            trace_kernel = "pb::k_trace<false"
            if default_cfg and world == 1 and not args.no_live_pmc:
                pmc, pmc_note = pmc_live(args.workload, spp, trace_kernel, args.pmc_budget)
            if pmc is None and default_cfg:
                live_note = pmc_note
                pmc, pmc_note = pmc_record(args.workload, spp, world, trace_kernel)
                pmc_note = f"{pmc_note} [committed record; {live_note}]" if pmc else f"{pmc_note}; {live_note}"
            traffic = frac_hbm_counter = valu = gather = None
            bound, fracs = "hbm", {}
            if pmc:
                traffic = pmc["hbm_bytes_per_dispatch_fetch_x2"]
                n_disp = max(pmc["dispatches"], 1)
                # HBM-side bytes per launch / the launch's duration when it has the GPU to itself / the 8 TB/s peak
                frac_hbm_counter = traffic / solo_launch_s / 1e9 / HBM_PEAK_GBS
                fracs["hbm"] = frac_hbm_counter
                # the kernel's own cycles: GRBM_GUI_ACTIVE is summed over the 8 XCDs (and over the launches of the pass)
                cycles = pmc.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
                clock_hz = cycles / n_disp / solo_launch_s if cycles else None   # (counter pass / solo timing of the same launches: an estimate)
                isa = isa_histogram({"c4": "k_trace<false, true, true, false>", "c5": "k_trace<false, true, true, false>"}.get(args.workload, "k_trace<false, false, true, false>"))
                if pmc.get("SQ_INSTS_VALU") and cycles and isa:
                    # VALU ceiling from the kernel's own instruction mix (round 6): wave-instructions issued x the mean measured cost of an instruction of
                    # the kernel's main loop, over 1024 SIMDs x the kernel's cycles.  Scalar instructions share the issue slot (~1.7 cycles each between
                    # vector ones, measured): reported as `with_salu`, a second estimate.  An estimator that reads above 1 is a broken model: null + why.
                    busy = pmc["SQ_INSTS_VALU"] * isa["mean_cycles_per_valu"] / (SIMDS * cycles)
                    busy_salu = (pmc["SQ_INSTS_VALU"] * isa["mean_cycles_per_valu"] + pmc.get("SQ_INSTS_SALU", 0.0) * 1.7) / (SIMDS * cycles)
                    valu = {"frac": busy if busy <= 1.0 else None,
                            "reason": None if busy <= 1.0 else f"the ISA-histogram model reads {busy:.2f} > 1: the static mix of the loop does not describe this launch's dynamic mix",
                            "with_salu": busy_salu if busy_salu <= 1.0 else None,
                            "mean_cycles_per_valu_inst": isa["mean_cycles_per_valu"], "full_rate_share": isa["full_rate_share"], "loop": {k: isa[k] for k in ("kernel", "valu_in_loop", "salu_in_loop", "vmem_in_loop", "top")},
                            "insts_valu_per_launch": pmc.get("SQ_INSTS_VALU", 0.0) / n_disp, "insts_salu_per_launch": pmc.get("SQ_INSTS_SALU", 0) / n_disp,
                            "active_cycles_per_launch": cycles / n_disp, "clock_ghz": clock_hz / 1e9 if clock_hz else None,
                            "lanes_per_valu": pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_INSTS_VALU"] if pmc.get("SQ_THREAD_CYCLES_VALU") and pmc.get("SQ_INSTS_VALU") else None,
                            "wait_any_frac": pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_WAVE_CYCLES") else None,
                            "counter_ratio_r5": pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cycles) if pmc.get("SQ_ACTIVE_INST_VALU") else None,
                            "note": "frac = SQ_INSTS_VALU x mean measured cycles per VALU instruction of the kernel's main loop (static histogram of the code object: "
                                    "full-rate fp32 add / mul / fma and simple integer / move / select 2.33 cycles, packed fp32, min / max, compares, conversions, shifts, "
                                    "integer multiplies 4.2, transcendentals 8.2 -- scripts/ubench/valu_rate*.hip) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); reads 1.0 on the "
                                    "VALU-bound calibration kernels by construction (u_fma 2.33, u_pk_fma 4.18, u_rcp 8.18 cycles per instruction: profiles/r5_ubench_pmc.txt); "
                                    "a static mix stands in for the dynamic one: an estimate; counter_ratio_r5 = round 5's SQ_ACTIVE_INST_VALU x 4 ratio, kept for comparison only"}
                    if busy <= 1.0:
                        fracs["valu"] = busy
                if pmc.get("TCP_TOTAL_ACCESSES_sum") and cycles:
                    # the vector-memory load path: 16-byte lane accesses x 16 B over the kernel's cycles, against what the same access
                    # pattern gets from an L1-resident table (GATHER_L1_BYTES_PER_CLK_CU, measured).  An estimate: reported, never `bound`.
                    b_clk_cu = pmc["TCP_TOTAL_ACCESSES_sum"] * 16.0 / (cycles * CUS)
                    tag, miss = pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0), pmc.get("TCP_TCC_READ_REQ_sum", 0.0)
                    # the peak for THIS kernel's items (round 6): nodes and curve records are 64-byte items (4 loads), the triangle leaves of a
                    # triangle-only scene 80-byte items (5 loads: TriPair), 48-byte triangle slots are priced as 64; the byte-weighted harmonic mean
                    nb = node_b * (sst["closest_nodes"] + sst["shadow_nodes"]) + 64.0 * (sst["closest_curves"] + sst["shadow_curves"]) / 2.0
                    tb = (80.0 / 2.0 if not desc.num_segments() else 64.0) * (sst["closest_tris"] + sst["shadow_tris"])
                    peak_items = (nb + tb) / (nb / GATHER_PEAK_BY_ITEM[64] + tb / GATHER_PEAK_BY_ITEM[80 if not desc.num_segments() else 64]) if (nb + tb) else GATHER_L1_BYTES_PER_CLK_CU
                    g_frac = b_clk_cu / peak_items
                    gather = {"frac": g_frac if g_frac <= 1.0 else None,
                              "reason": None if g_frac <= 1.0 else f"reads {g_frac:.2f} > 1 of the micro-benchmark's rate for this item mix: the kernel's accesses share lines (lanes_per_line) the benchmark's do not",
                              "bytes_per_clk_cu": b_clk_cu, "peak_bytes_per_clk_cu": peak_items,
                              "l1_hit_rate": 1.0 - miss / tag if tag else None, "lanes_per_line": pmc["TCP_TOTAL_ACCESSES_sum"] / tag if tag else None,
                              "l2_read_latency_cycles": pmc["TCP_TCC_READ_REQ_LATENCY_sum"] / miss if miss and pmc.get("TCP_TCC_READ_REQ_LATENCY_sum") else None,
                              "note": "ESTIMATE (does not set `bound`): TCP_TOTAL_ACCESSES x 16 B per clock and CU over the measured rate of per-lane 64-byte gathers "
                                      "from an L1-resident table, by item footprint (scripts/ubench/vmem_gather2.hip: 64-byte items 23.9, 80-byte items 28.0 B/clk/CU at 1-6 blocks per CU; L2-resident 21.9; "
                                      "coalesced control 31.7; beyond L2 the unit of cost is the 128-byte line: ~60 G lines/s chip-wide)"}
                ok = {k: v for k, v in fracs.items() if v is not None and v <= 1.0}   # a ceiling fraction above 1 is a broken model, never a bound
                if ok:
                    bound = max(ok, key=ok.get)
            roofline = {"bound": bound, "kernel": "k_trace (closest-hit rays of bounce k + shadow rays of bounce k-1)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "note": "achieved / frac = ALGORITHMIC bytes (64 B per node visit -- binary tree or the Q tree's quantised 4-wide node --, 48 B per triangle test, 32 B per curve-piece test on the Q tree (64 on the binary tree), 52 B per ray: SURVEY 8d) / kernel "
                                "time (HIP events on the launch's own stream, timed region) -- NOT a ceiling: the scene is served from L2 / "
                                "Infinity Cache, so it can exceed 1.  The ceilings are frac_hbm_counter (HBM-side bytes from the FETCH_SIZE / "
                                "WRITE_SIZE PMC passes per launch / solo launch duration / 8 TB/s) and valu.frac (VALU pipes busy, a counter ratio); "
                                "gather.frac (the vector L1's per-lane gather rate) is an estimate.  By default the frame runs as two path groups on two HIP streams whose launches overlap, so "
                                "per-launch durations of the timed region include shared time; `solo` and kernel_ms_per_step are from an "
                                "untimed extra render as ONE group, every launch alone on the GPU.  `bound` names the largest COUNTER-DERIVED fraction "
                                "(frac_hbm_counter, valu.frac) that is <= 1; when none of them is near 1 (see `limiter`) the kernel is bound by the latency "
                                "of its dependent item fetches times the rays in flight, not by a unit's throughput",
                        "limiter": (None if not fracs else ("latency of dependent fetches x rays in flight: no unit above 0.7 (" + ", ".join(f"{k} {v:.2f}" for k, v in sorted(fracs.items())) + ")"
                                                            if max(fracs.values()) < 0.7 else max(fracs, key=fracs.get))),
                        "solo": {"launches_per_step": solo["n_trace_closest"], "avg_launch_ms": solo_launch_s * 1e3,
                                 "achieved": bytes_step / (solo_ms * 1e-3) / 1e9,
                                 "frac": bytes_step / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_frame": solo["ms_total"]},
                        "traffic": traffic, "frac_hbm_counter": frac_hbm_counter, "valu": valu, "gather": gather, "pmc_source": pmc_note,
                        "algorithmic_bytes_per_launch": bytes_step / max(launches, 1), "node_bytes": node_b, "curve_bytes": curve_b,
                        "avg_launch_ms": ms_step / max(launches, 1), "launches_per_step": launches,
                        "rays_per_step": sst["closest_rays"] + sst["shadow_rays"],
                        "kernel_ms_per_step": {{"trace_closest": "trace", "surface": "classify"}.get(k[3:], k[3:]): solo[k]
                                               for k in solo if k.startswith("ms_") and k not in ("ms_total",)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload) if not args.spp else None

    if rank == 0:
        samples = W * H * spp * args.steps
        out = {
            "metric": "Msamples/s (paths x spp / s), 1920x1080 Cornell-box-Suzanne render", "value": samples / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["desc"] + (f" x {world} (spp scaled with the GPU count)" if (world > 1 and weak) else ""), "width": W, "height": H, "spp": spp,
                       "exchange": None if world == 1 else {"gather": "pbrhip_comm_gather_layer (RCCL send/recv of the ranks' shards)",
                                                            "reduce": "pbrhip_comm_reduce_layer (ncclReduce f32 + u32)",
                                                            "torch": "torch.distributed.reduce"}[exchange],
                       "triangles": desc.num_triangles(), "curve_segments": desc.num_segments(),
                       "bvh_nodes": info["num_nodes"], "bvh_depth": info["depth"], "scene_bytes": info["device_bytes"],
                       "parallelism": f"{shard_block}x{shard_block} pixel blocks, block index % {world}" if world > 1 else "1gpu",
                       "rng": "PCG32((pass<<32)+pixel, 1234567890)"},
            "roofline": roofline, "cpu_baseline": cpu, "host_layer": host_layer,
            "metric_definition": {"version": 2,
                                  "value": "a step = Render() of the frame with the scene resident in HBM; it ends with the complete RenderLayer (rgba f32 + count u32) "
                                           "in the HBM of rank 0 (N > 1: after the exchange over xGMI)",
                                  "why": "the run's measurement contract: \"`value` is whole-job throughput with inputs already resident in HBM when the timed region starts "
                                         "(if the boundary hands over host buffers, note the PCIe-inclusive rate in DESIGN.md -- it is never `value`)\"; pbrlab's "
                                         "RenderLayer is host memory, so the PCIe-inclusive figure is reported beside it as host_layer.  Rounds 1-4 (version 1) reported the "
                                         "host-inclusive figure as `value`: compare like with like -- r4 value with r5 / r6 host_layer.value",
                                  "host_copy": "41.5 MB (1920 x 1080 x 20 B) over PCIe in 0.7-0.8 ms = its ~55 GB/s: the copy cannot start before the last pass is "
                                               "accumulated (the layer is a sum over all passes) -- it is the floor of host_layer.ms_per_step - ms_per_step"},
        }
        if diag is not None:   # N > 1: where each rank's time went (gathered after the timed region)
            out.update(diag)
        if other is not None:
            out["weak" if not weak else "strong"] = other
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
