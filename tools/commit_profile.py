#!/usr/bin/env python3
"""Copies a tools/profile.sh run into profiles/<round>/<tag>_* and refreshes profiles/traffic.json from its summary:
   python tools/commit_profile.py <gpurun tag> <round dir> <tag> [bench json]
e.g.  python tools/commit_profile.py r5a r05 a gpurun_out/r5_a_bench.json"""
import json, os, shutil, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, rnd, tag = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(R, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
summary = os.path.join(R, "gpurun_out", src + "_summary.txt")
shutil.copy(summary, os.path.join(dst, tag + "_summary.txt"))
shutil.copy(os.path.join(R, "gpurun_out", src + "_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
if len(sys.argv) > 4:
    shutil.copy(sys.argv[4], os.path.join(dst, tag + "_bench.json"))
line = [l for l in open(summary) if l.startswith("TRAFFIC_JSON ")][-1]
out = {"source": "profiles/%s/%s_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 3 --warmup 1`; bytes = "
                 "(2 x FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction)" % (rnd, tag),
       "objects": 10000000, "bytes_per_launch": json.loads(line[len("TRAFFIC_JSON "):])}
json.dump(out, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
print("profiles/%s/%s_* written, traffic.json refreshed" % (rnd, tag))
