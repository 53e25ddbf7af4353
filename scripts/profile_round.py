#!/usr/bin/env python3
"""Collects the round's profile evidence for one workload on the GPU box and writes it under profiles/ (via gpurun_out/):

  r6_<wl>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py --workload <wl> --steps 3 --warmup 1`
  r6_<wl>_pmc.json           per kernel: FETCH_SIZE, WRITE_SIZE (HBM-side traffic, separate passes as MI355X_MICROARCH.md
                             prescribes; gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2), SQ instruction / wait
                             counters and TCC hit / miss / EA read requests, over ONE render of the workload
                             (scripts/render_once.py, one path group: every launch alone on the GPU), plus csrc_hash --
                             bench.py only uses the record while the kernel sources are the ones it was taken on.

usage (GPU box):  python scripts/profile_round.py c2|c3|c4 [--no-stats]      -> gpurun_out/profiles/..."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_hash, WORKLOADS; importing it does not touch the GPU)

PASSES = ["FETCH_SIZE", "WRITE_SIZE",
          "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE",
          "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum",
          "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum",
          "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"]  # the vector-memory gather path: address / data-return units busy, over the kernels' active cycles


def main():
    wl = sys.argv[1]
    w = bench.WORKLOADS[wl]
    variant = {"c2": "ggx", "c3": "sss", "c4": "hair"}[wl]
    out = os.path.join(ROOT, "gpurun_out", "profiles")
    tmp = os.path.join(ROOT, "gpurun_out", f"prof_tmp_{wl}")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", VARIANT=variant, SPP=str(w["spp"]), PBRHIP_STREAMS="1", REPS="1")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for i, counters in enumerate(PASSES):
        d = os.path.join(tmp, f"p{i}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + counters.split() + ["-f", "csv", "-d", d, "-o", f"p{i}", "--", "python3", os.path.join(ROOT, "scripts", "render_once.py")]
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=900)
        print(f"pass {counters.split()[0]}...: rc {r.returncode}", flush=True)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if i == 0:
                    disp[k].add(row["Dispatch_Id"])
        shutil.rmtree(d, ignore_errors=True)
    kernels = {}
    for k, v in sorted(agg.items()):
        if not k.startswith("pb::"):
            continue
        n = max(len(disp[k]), 1)
        rec = {"dispatches": len(disp[k])}
        rec.update({c: x for c, x in v.items()})
        fetch, write = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        rec["hbm_bytes_per_dispatch_raw"] = (fetch + write) * 1024 / n
        rec["hbm_bytes_per_dispatch_fetch_x2"] = (2 * fetch + write) * 1024 / n
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            rec["tcc_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        if v.get("SQ_INSTS_VALU"):
            rec["lanes_per_valu"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / v["SQ_INSTS_VALU"]
        kernels[k] = rec
    json.dump({"csrc_hash": bench.csrc_hash(), "workload": wl, "spp": w["spp"], "kernels": kernels,
               "source": f"scripts/profile_round.py {wl}: rocprofv3 --pmc passes {[p.split()[0] for p in PASSES]} over one 1920x1080 x "
                         f"{w['spp']} spp render (one path group); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 on gfx950"},
              open(os.path.join(out, f"r6_{wl}_pmc.json"), "w"), indent=1, sort_keys=True)
    for k, rec in kernels.items():
        print(f"{k[:44]:44s} n={rec['dispatches']:4d} hbm/launch {rec['hbm_bytes_per_dispatch_fetch_x2'] / 1e9:7.3f} GB  valu {rec.get('SQ_INSTS_VALU', 0):.3g} "
              f"lanes {rec.get('lanes_per_valu', 0):.1f}  tcc hit {rec.get('tcc_hit_rate', 0):.3f}")
    if "--no-stats" not in sys.argv:
        d = os.path.join(tmp, "stats")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "-f", "csv", "-d", d, "-o", "ks", "--", "python3", os.path.join(ROOT, "bench.py"),
               "--workload", wl, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc"]  # (no profiler inside the profiler)
        r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", capture_output=True, text=True, timeout=1500)
        print("kernel-trace: rc", r.returncode, r.stdout.strip().splitlines()[-1][:300] if r.stdout.strip() else r.stderr[-500:])
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            shutil.copy(f, os.path.join(out, f"r6_{wl}_kernel_stats.csv"))
        if r.stdout.strip():
            open(os.path.join(out, f"r6_{wl}_bench_under_rocprof.json"), "w").write(r.stdout.strip().splitlines()[-1] + "\n")
        shutil.rmtree(d, ignore_errors=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
