"""Times the expanding scan ALONE on a sorted 10 M tree (one extend + par_sort, then the scan kernel over and over through
bp_layer_scan_shard_pairs_device: no pair sort behind it, its output never consumed) -- for library variants whose scan output may be
WRONG (timing experiments, tools/ab.sh build <v> "-DSX_EXP=...").
   python tools/sx_micro.py [n_objects] [variant ...]      (variant = a directory of broadphase_rs_amd/lib_ab, or "base")
Each variant runs in its own process (the library is chosen by BP_LIBRARY at import)."""
import os, subprocess, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if os.environ.get("SX_MICRO_CHILD"):
    sys.path.insert(0, R)
    import numpy as np
    import torch
    import broadphase_rs_amd as bp
    from broadphase_rs_amd import scenes
    n = int(sys.argv[1])
    dev = torch.device("cuda:0")
    sysb, bounds, ids = scenes.gen_boxes_powerlaw(n, seed=0, alpha=2.5) if os.environ.get("SX_POWERLAW") else scenes.gen_boxes(n, seed=1)
    d_bounds = torch.from_numpy(bounds).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    L = bp.LayerBuilder.new().with_min_depth(0).with_device(0).build(bp.Index64_3D)
    # (the third frame's sort takes the compact route: depth / count words for the scan.  The warm-up frames do NOT scan: a variant
    # whose scan output is wrong must not feed the pair sort, whose kernels trust what they are given)
    for _ in range(5):
        L.clear(); L.extend(sysb, d_bounds, d_ids); L.par_sort()
    for _ in range(3): out, uniq = L.scan_shard_pairs_device(0, 1)
    L.profile_enable(True); L.profile_reset()
    K = 10
    for _ in range(K): out, uniq = L.scan_shard_pairs_device(0, 1)
    ms, cnt = L.profile()["scan_emit"]
    st = L.stats()
    print("%-8s scan %.1f us per launch (%d launches)  raw pairs %d  compact sorts %d" % (os.environ.get("SX_MICRO_NAME"), 1e3 * ms / max(cnt, 1), cnt, len(out), st.get("tree_sorts_compact", -1)))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].isdigit() else "10000000"
variants = [v for v in sys.argv[1:] if not v.isdigit()] or ["base"]
for v in variants:
    lib = os.path.join(R, "broadphase_rs_amd", "lib" if v == "base" else os.path.join("lib_ab", v), "libbroadphase_amd.so")
    env = dict(os.environ, SX_MICRO_CHILD="1", SX_MICRO_NAME=v, BP_LIBRARY=os.path.abspath(lib))
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), n], env=env, capture_output=True, text=True, timeout=240)
        print((r.stdout.strip().splitlines() or ["%s: no output (rc %d) %s" % (v, r.returncode, r.stderr.strip()[-300:])])[-1], flush=True)
    except subprocess.TimeoutExpired:
        print(v, "timed out", flush=True)
