#!/usr/bin/env python3
"""Ad-hoc PMC passes over one render (scripts/render_once.py):  python scripts/pmc_adhoc.py c2 "CTR_A CTR_B" "CTR_C" ...
Prints, per kernel, the summed counters (one rocprofv3 --pmc run per argument; no trace flags next to --pmc)."""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = sys.argv[1]
w = bench.WORKLOADS[wl]
variant = {"c2": "ggx", "c3": "sss", "c4": "hair"}[wl]
spp = os.environ.get("SPP", str(w["spp"]))
env = dict(os.environ, TMPDIR="/tmp", VARIANT=variant, SPP=spp, PBRHIP_STREAMS="1", REPS="1")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for i, counters in enumerate(sys.argv[2:]):
    d = f"/tmp/pmc_adhoc_{i}"
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters.split() + ["-f", "csv", "-d", d, "-o", f"p{i}", "--", "python3", os.path.join(ROOT, "scripts", "render_once.py")]
    try:
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=int(os.environ.get("PASS_TIMEOUT", "150")))
    except subprocess.TimeoutExpired:
        print(f"pass {counters}: timed out", flush=True)
        continue
    print(f"pass {counters}: rc {r.returncode}", flush=True)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    shutil.rmtree(d, ignore_errors=True)
for k, v in sorted(agg.items()):
    if k.startswith("pb::"):
        print(k, json.dumps({c: x for c, x in sorted(v.items())}))
