import json, sys
v = sys.argv[1]
try:
    d = json.load(open("gpurun_out/ab_%s.json" % v))
except Exception as e:  # noqa: BLE001
    print(v, "no result:", e); sys.exit(0)
k = d["kernels"]
print("%-14s step %.3f cold %.3f | " % (v, d["ms_per_step"], d["cold_frame_ms"]) +
      " ".join("%s %.3f" % (c, k[c]["ms_per_step"]) for c in ("encode", "tree_sort", "tree_finish", "scan_emit", "pair_sort", "dedup")) +
      " | pairs %d parity_hash %s" % (d["config"]["pairs"], d.get("list_hash")))
