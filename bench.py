#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Layer extend -> sort -> scan path on MI355X.

Metric (BASELINE.json): candidate pairs/sec at 10M Index64_3D AABBs; achieved HBM GB/s vs peak.
A "step" is one pass of the hot path over one batch of synthetic AABBs that are already resident
in HBM: clear + extend + par_sort + par_scan, result (sorted, duplicate-free pair list) left in HBM.
value = unique pairs produced by all ranks / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--objects M] [--scaling strong|weak]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU,
RCCL).  Default for N > 1 is BASELINE.json configs[3] as stated: `--objects` (10 M) objects IN TOTAL, sharded by
object index over the N ranks ("scaling": "strong"); `--scaling weak` gives every rank `--objects` objects of one
N-times larger scene.  Entries and pairs are exchanged by Morton / id range (broadphase_rs_amd/dist.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
M64 = (1 << 64) - 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--objects", type=int, default=10_000_000,
                    help="objects of the scene (per GPU with --scaling weak, in total with --scaling strong)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 only: strong = configs[3] (--objects in total), weak = --objects per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-objects", type=int, default=10_000_000, help="size of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1m", action="store_true", help="skip the configs[1] (1 M objects) sub-record")
    ap.add_argument("--no-moving", action="store_true", help="skip the moving-scene sub-record")
    ap.add_argument("--no-extra", action="store_true", help="skip the mixed_scale / aabb_filter / one_shot sub-records")
    ap.add_argument("--mixed-objects", type=int, default=12_500_000, help="objects of the mixed_scale sub-record (configs[4]'s per-GPU share)")
    ap.add_argument("--moving-frames", type=int, default=20)
    ap.add_argument("--sharded", action="store_true", help="run the multi-GPU pipeline even with one rank (rehearsal)")
    ap.add_argument("--pair-order", choices=["global", "local"], default="local",
                    help="N > 1: 'local' (default, north_star's design: boundary keys are exchanged, every rank returns the sorted "
                         "duplicate-free pairs of its Morton range; the union is the reference's set) or 'global' (pairs exchanged by "
                         "first-id range: the ranks' lists concatenate to the reference's sorted list).  The other order is timed as a sub-record")
    ap.add_argument("--one-order", action="store_true", help="N > 1: skip the sub-record of the other pair order")
    ap.add_argument("--aabb-filter", action="store_true",
                    help="variant (not the headline metric): par_scan_filtered with the built-in device AABB narrow phase")
    return ap.parse_args()


def usable_cpus(hw):
    """threads this process may really use: affinity mask and cgroup quota, not the host's core count"""
    n = hw
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def list_hash_numpy(pairs):
    """order-sensitive 64-bit fingerprint of a pair list: (count, sum p, sum p * (2 i + 1)) mod 2^64, p = id << 32 | id_"""
    import numpy as np
    p = (pairs[:, 0].astype(np.uint64) << np.uint64(32)) | pairs[:, 1].astype(np.uint64)
    w = np.arange(len(p), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with np.errstate(over="ignore"):
        return [int(len(p)), int(p.sum(dtype=np.uint64)), int((p * w).sum(dtype=np.uint64))]


def set_hash_numpy(pairs):
    """order-INDEPENDENT fingerprint of a pair set: (count, sum p, sum p * p) mod 2^64"""
    import numpy as np
    p = (pairs[:, 0].astype(np.uint64) << np.uint64(32)) | pairs[:, 1].astype(np.uint64)
    with np.errstate(over="ignore"):
        return [int(len(p)), int(p.sum(dtype=np.uint64)), int((p * p).sum(dtype=np.uint64))]


def list_hash_torch(t):
    """the same fingerprint of a device-resident (n, 2) int32 view, with torch ops (int64 arithmetic wraps mod 2^64)"""
    import torch
    a = t[:, 0].to(torch.int64) & 0xFFFFFFFF
    b = t[:, 1].to(torch.int64) & 0xFFFFFFFF
    p = (a << 32) | b
    w = torch.arange(p.numel(), dtype=torch.int64, device=p.device) * 2 + 1
    return [int(p.numel()), int(p.sum().item()) & M64, int((p * w).sum().item()) & M64]


def set_hash_torch(t):
    import torch
    a = t[:, 0].to(torch.int64) & 0xFFFFFFFF
    b = t[:, 1].to(torch.int64) & 0xFFFFFFFF
    p = (a << 32) | b
    return [int(p.numel()), int(p.sum().item()) & M64, int((p * p).sum().item()) & M64]


def cpu_baseline(n_objects, seed):
    """The oracle (C restatement of the reference's algorithm and structure) timed on this host.
    kind = "port": the Rust crate itself cannot be built here (no rustc/cargo).  Bounded sample:
    the same gen_boxes recipe and density at n_objects objects, full extend+par_sort+par_scan."""
    import oracle  # checker only: never on the measured GPU path
    from broadphase_rs_amd import scenes
    sysb, bounds, ids = scenes.gen_boxes(n_objects, seed=seed)
    threads = usable_cpus(oracle.lib().orc_hw_threads())
    rows = {}
    list_hash = None
    set_hash = None
    # "tuned": min_depth = smallest natural depth of the scene, so indices/results are unchanged
    # but par_scan_impl can actually split (src/layer.rs:533-546); "default": min_depth = 0, where
    # level_mask(0) == 0 makes the reference's scan one serial sweep (only the sorts are parallel).
    probe = oracle.OracleLayer(oracle.INDEX64_3D, 0)
    probe.extend(sysb, bounds[:20000], ids[:20000])
    natural = int((probe.keys & 31).min())
    for name, md in (("default_min_depth_0", 0), ("tuned_min_depth_%d" % natural, natural)):
        lay = oracle.OracleLayer(oracle.INDEX64_3D, md)
        t0 = time.perf_counter()
        lay.clear()
        lay.extend(sysb, bounds, ids)
        lay.par_sort(threads)
        pairs = lay.par_scan(threads)
        dt = time.perf_counter() - t0
        rows[name] = {"pairs": int(len(pairs)), "seconds": dt, "pairs_per_s": len(pairs) / dt}
        if md == 0:
            list_hash = list_hash_numpy(pairs)
            set_hash = set_hash_numpy(pairs)
        del lay, pairs
    best = max(rows.values(), key=lambda r: r["pairs_per_s"])
    return {"value": best["pairs_per_s"] / 1e6, "unit": "Mpairs/s", "cores": threads, "kind": "port",
            "sample": "gen_boxes(%d objects, density 0.001, seed %d): clear+extend+par_sort+par_scan, OpenMP "
                      "C restatement with the reference's structure; rows: %s" % (
                          n_objects, seed, json.dumps({k: round(v["pairs_per_s"] / 1e6, 3) for k, v in rows.items()})),
            "rows": rows, "list_hash": list_hash, "set_hash": set_hash}


# one kernel per class on the default path (bp_kernel_class of include/bp_layer.h) -> (kernel name, SURVEY 8(d) bytes per launch)
EVENT_EVERY = 4   # the timed region's HIP events bracket the dominant class in every 4th frame (run_single)


def kernel_table(n, M, P_raw, P, cells=False, tree_cells=False, tree_compact=False):
    """cells / tree_cells: the timed frames' pair / tree sorts took the cell route (bp_stats.sorts_in_cells, tree_sorts_in_cells:
    two claiming passes, no stable pass); tree_compact: the tree went through encode -> passes -> finish as 8-byte compact records
    (bp_stats.tree_sorts_compact) -- the tree classes' bytes are then what these kernels move (less than SURVEY 8(d)'s 12-byte entries)"""
    if tree_compact:
        tree_sort = ("radix_claim_kernel<NoVal, false, 1> + radix_claim_kernel<NoVal, true, 1> (tree-sort passes on compact 8-byte records: index bits | "
                     "id | box code in one word; both claim whole aligned lines -- by the high digit, then every high bin into cells laid out "
                     "from the previous sort's group sizes; no stable pass)")
    elif tree_cells:
        tree_sort = ("radix_claim_kernel<u32, false, 0> + radix_claim_kernel<u32, true, 0> (tree-sort passes: both claim whole aligned lines -- by the "
                     "high digit, then every high bin into cells laid out from the previous sort's group sizes; no stable pass)")
    else:
        tree_sort = ("radix_claim_kernel<u32, false, 0> + radix_onesweep_kernel<u64,u32> (tree-sort passes: the first claims whole aligned lines, "
                     "the second is the stable one-sweep pass)")
    pair_sort = ("radix_claim_kernel<NoVal, false, 0> + radix_claim_kernel<NoVal, true, 0> (pair-sort passes: both claim whole aligned lines -- by the "
                 "high digit, then every high bin into cells laid out from the previous frame's group sizes; no stable pass)") if cells else \
                ("radix_claim_kernel<NoVal, false, 0> + radix_onesweep_kernel<u64,NoVal> (pair-sort passes: the first claims whole aligned lines, "
                 "the second is the stable one-sweep pass)")
    return {
        "encode": ("encode_count_kernel + encode_emit_kernel (contains + to_local + indices: AABB -> index entries)",
                   28.0 * n + (8.0 if tree_compact else 12.0) * M),
        "tree_sort": (tree_sort, (16.0 if tree_compact else 24.0) * M),
        "tree_finish": ("tgroup_finishc_kernel (LDS finish of the tree sort: compact records in; index values, {id, context} words and depth / "
                        "range-length words out)", 8.0 * M + 18.0 * M)
                       if tree_compact else ("tgroup_finish_kernel (LDS finish of the tree sort)", 24.0 * M),
        "scan_emit": ("scan_expand_kernel (overlap scan, single pass)", 12.0 * M + 8.0 * P_raw),
        "pair_sort": (pair_sort, 16.0 * P_raw),
        "dedup": ("group_finish_kernel (LDS finish of the pair sort + dedup)", 8.0 * P_raw + 8.0 * P),
    }


def run_single(bp, scenes, torch, np, dev, local_rank, n, seed, warmup, steps, aabb_filter=False, detail=True, events=True):
    """steady-state frames of one layer on one GPU -> dict of measurements (all times in ms).
    events: the dominant class is timed with HIP events inside the timed region (the roofline's duration) -- in every
    EVENT_EVERY-th frame (every 2nd of a short run): an event pair costs ~5 us of stream bubble, four of them per frame would be 1 % of the 10 M step
    and 5 % of the 1 M frame.  A run that reports no roofline (events=False) records none."""
    sysb, bounds, ids = scenes.gen_boxes(n, seed=seed)
    d_bounds = torch.from_numpy(bounds).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    layer = bp.LayerBuilder.new().with_min_depth(0).with_device(local_rank).build(bp.Index64_3D)

    def step():
        layer.clear()
        layer.extend(sysb, d_bounds, d_ids)
        layer.par_sort()
        if aabb_filter:
            return layer.scan_filtered_aabb_device(d_bounds)
        return layer.scan_device()

    def timed(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            out = step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3, out

    res = {}
    res["first_frame_ms"], _ = timed(1)            # fresh layer: includes every hipMalloc
    layer.reset_plan_hints()
    res["cold_frame_ms"], _ = timed(1)             # buffers allocated, no plan hints: what a layer's first frame runs
    st = layer.stats()
    res["cold_plan"] = {"tree_sort_passes": st["tree_sort_passes"], "pair_sort_passes": st["pair_sort_passes"]}
    # warm-up with every class timed: finds the dominant kernel = largest TOTAL time per step over all classes
    layer.profile_reset(); layer.profile_select(None); layer.profile_enable(True)
    for _ in range(max(warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    layer.profile_enable(False)
    wprof = layer.profile()
    st = layer.stats()
    table = kernel_table(n, st["entries"], st["raw_pairs"], st["pairs"])
    dom_cls = max(table, key=lambda c: wprof[c][0])
    cells0, tcells0, tcomp0 = layer.stats()["sorts_in_cells"], layer.stats()["tree_sorts_in_cells"], layer.stats()["tree_sorts_compact"]
    # timed region: events only around the dominant class (every event pair costs a few us of stream bubble)
    layer.profile_reset(); layer.profile_select([dom_cls]); layer.profile_enable(events)
    every = EVENT_EVERY if steps >= 4 * EVENT_EVERY else 2 if steps >= 4 else 1   # (at least two timed frames)
    layer.profile_sample(every)
    res["event_every"] = every if events else 0
    syncs0 = layer.stats()["host_syncs"]
    res["ms_per_step"], out = timed(steps)
    layer.profile_sample(1)
    res["host_syncs_per_step"] = (layer.stats()["host_syncs"] - syncs0) / steps
    res["cell_route"] = (layer.stats()["sorts_in_cells"] - cells0) == steps   # every timed frame's pair sort took the cell route
    res["tree_cell_route"] = (layer.stats()["tree_sorts_in_cells"] - tcells0) == steps
    res["tree_compact_route"] = (layer.stats()["tree_sorts_compact"] - tcomp0) == steps
    layer.profile_enable(False)
    res["dom_cls"], res["dom_prof"] = dom_cls, layer.profile()[dom_cls]
    if detail:
        k = max(1, min(steps, 3))
        layer.profile_reset(); layer.profile_select(None); layer.profile_enable(True)
        for _ in range(k):
            out = step()
        torch.cuda.synchronize()
        layer.profile_enable(False)
        res["detail_steps"], res["prof"] = k, layer.profile()
    res["stats"], res["n_pairs"], res["out"], res["layer"] = layer.stats(), len(out), out, layer
    res["scene"], res["d_bounds"] = (sysb, d_ids), d_bounds
    return res


def moving_frame(torch, base, size, side, frame, seed):
    """frame `frame` of the moving scene: every box displaced by a seeded jitter of up to +-0.5 of its own size per axis
    (relative to its rest position, so the density stays the recipe's), kept inside the system bounds"""
    g = torch.Generator(device=base.device)
    g.manual_seed(seed * 1000003 + frame)
    jit = (torch.rand((base.shape[0], 3), generator=g, device=base.device, dtype=torch.float32) - 0.5) * size
    mn = torch.minimum(torch.clamp(base[:, :3] + jit, min=0.0), side - size)
    return torch.cat([mn, mn + size], dim=1).contiguous()


def run_moving(bp, scenes, torch, np, dev, local_rank, n, seed, frames, check_last):
    """the reference's frame loop moves its objects (examples/main.rs:410-418, 508-511): the same n objects, re-placed every
    frame -> entries per object, group sizes and digit plans drift, so the plan hints of the previous frame are predictions
    again.  The displacement runs on the device OUTSIDE the timed region (torch ops); every frame is timed on its own."""
    sysb, bounds, ids = scenes.gen_boxes(n, seed=seed)
    base = torch.from_numpy(bounds).to(dev)
    size = (base[:, 3:] - base[:, :3]).contiguous()
    side = float(sysb[1][0])
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    layer = bp.LayerBuilder.new().with_min_depth(0).with_device(local_rank).build(bp.Index64_3D)

    def step(cur):
        layer.clear()
        layer.extend(sysb, cur, d_ids)
        layer.par_sort()
        return layer.scan_device()

    for f in range(2):                                   # the layer's first frames (allocations, plan probing) are not the subject
        out = step(moving_frame(torch, base, size, side, f, seed))
    torch.cuda.synchronize()
    st0 = layer.stats()
    times, pairs = [], []
    cur = None
    for f in range(2, 2 + frames):
        cur = moving_frame(torch, base, size, side, f, seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = step(cur)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        pairs.append(len(out))
    st1 = layer.stats()
    rec = {"workload": "the same %d objects, every frame re-placed by a seeded jitter of +-0.5 x its size per axis (device-side, outside "
                       "the timed region); %d frames, each timed on its own (clear+extend+par_sort+par_scan)" % (n, frames),
           "frames": frames, "ms_per_step": sum(times) / len(times), "ms_min": min(times), "ms_max": max(times),
           "ms_frames": [round(t, 4) for t in times],
           "pairs_mean": sum(pairs) / len(pairs), "value": sum(pairs) / (sum(times) * 1e-3), "unit": "pairs/s",
           "host_syncs_per_step": (st1["host_syncs"] - st0["host_syncs"]) / frames,
           "plan_events": {k: st1[k] - st0[k] for k in ("extend_reruns", "tree_plan_flips", "pair_plan_flips", "sort_repairs",
                                                        "sorts_ahead_missed")},
           "sorts_ahead": st1["sorts_ahead"] - st0["sorts_ahead"], "sorts_in_cells": st1["sorts_in_cells"] - st0["sorts_in_cells"],
           "tree_sorts_in_cells": st1["tree_sorts_in_cells"] - st0["tree_sorts_in_cells"],
           "plan_events_note": "extend_reruns: extends launched on the previous frame's sizes that had to run again; *_plan_flips: sorts "
                               "whose grouped-finish plan differs from the previous frame's; sort_repairs: deferred tree-sort checks that "
                               "failed (full-width re-sort + second scan); sorts_ahead: frames whose pair sort was enqueued behind the scan on the "
                               "previous frame's sizes (no host round trip in between), sorts_ahead_missed: those that outgrew the sizes "
                               "and ran again"}
    if check_last:                                       # the last frame against the CPU restatement on the same boxes
        import oracle  # checker only
        ref = oracle.OracleLayer(oracle.INDEX64_3D, 0)
        ref.extend(sysb, cur.cpu().numpy(), ids)
        threads = usable_cpus(oracle.lib().orc_hw_threads())
        ref.par_sort(threads)
        exp = ref.par_scan(threads)
        rec["last_frame_parity"] = list_hash_numpy(exp) == (list_hash_torch(out.as_torch()) if len(out) else [0, 0, 0])
    else:
        rec["last_frame_parity"] = None
    return rec


def run_mixed_scale(bp, scenes, torch, np, dev, local_rank, n, seed, steps, check):
    """configs[4]'s per-GPU share: n objects with Pareto(alpha = 2.5) sizes (scenes.gen_boxes_powerlaw) -- long ranges, the
    work-list kernel, a deep stack in the reference.  Steady-state frames like the headline; the pair SET is compared with
    the CPU restatement's on the same boxes (check) -- fingerprint (count, sum p, sum p^2 mod 2^64)."""
    sysb, bounds, ids = scenes.gen_boxes_powerlaw(n, seed=seed, alpha=2.5)
    d_bounds = torch.from_numpy(bounds).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    layer = bp.LayerBuilder.new().with_min_depth(0).with_device(local_rank).build(bp.Index64_3D)

    def step():
        layer.clear(); layer.extend(sysb, d_bounds, d_ids); layer.par_sort()
        return layer.scan_device()

    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = layer.stats()
    rec = {"workload": "configs[4]'s per-GPU share: %d Index64_3D AABBs, isotropic Pareto(x_m = 1, alpha = 2.5) sizes capped at L / 4, "
                       "gen_boxes placement and density; clear+extend+par_sort+par_scan, steady state, %d steps" % (n, steps),
           "ms_per_step": ms, "entries": st["entries"], "candidates_examined": st.get("raw_candidates", 0), "raw_pairs": st["raw_pairs"],
           "pairs": len(out), "value": len(out) / (ms * 1e-3), "unit": "pairs/s",
           "tree_sort_passes": st["tree_sort_passes"], "pair_sort_passes": st["pair_sort_passes"],
           "set_hash": set_hash_torch(out.as_torch()) if len(out) else [0, 0, 0]}
    if check:
        import oracle  # checker only
        ref = oracle.OracleLayer(oracle.INDEX64_3D, 0)
        ref.extend(sysb, bounds, ids)
        threads = usable_cpus(oracle.lib().orc_hw_threads())
        t0 = time.perf_counter()
        ref.par_sort(threads)
        exp = ref.par_scan(threads)
        rec["cpu_seconds"] = time.perf_counter() - t0
        rec["parity"] = set_hash_numpy(exp) == rec["set_hash"] and list_hash_numpy(exp) == list_hash_torch(out.as_torch())
        rec["parity_note"] = "pair list (count, sum p, sum p^2, sum p (2 i + 1) mod 2^64) == the CPU restatement's on the same boxes"
    else:
        rec["parity"] = None
    del layer
    return rec


def run_aabb_filter(bp, torch, np, dev, r, d_bounds, steps):
    """scan_filtered(|a, b| bounds[a] overlaps bounds[b]) with the predicate inside the fused scan, on the
    headline scene and the headline's layer.  Parity: the same predicate applied with torch ops to the UNFILTERED device list
    (itself checked against the CPU list by parity_in_run) must leave the same list."""
    layer = r["layer"]
    sysb, d_ids = r["scene"]

    def step():
        layer.clear(); layer.extend(sysb, d_bounds, d_ids); layer.par_sort()
        return layer.scan_filtered_aabb_device(d_bounds)

    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    got = list_hash_torch(out.as_torch()) if len(out) else [0, 0, 0]
    n_out = len(out)
    # the reference filter on the unfiltered list (inclusive overlap, geom.rs:112-119), in slices to bound the temporaries
    layer.clear(); layer.extend(sysb, d_bounds, d_ids); layer.par_sort()
    full = layer.scan_device().as_torch()
    keep = torch.empty(full.shape[0], dtype=torch.bool, device=full.device)
    CH = 1 << 24
    for a in range(0, full.shape[0], CH):
        ia = full[a:a + CH, 0].to(torch.int64) & 0xFFFFFFFF
        ib = full[a:a + CH, 1].to(torch.int64) & 0xFFFFFFFF
        ba, bb = d_bounds[ia], d_bounds[ib]
        keep[a:a + CH] = ((ba[:, :3] <= bb[:, 3:]) & (bb[:, :3] <= ba[:, 3:])).all(dim=1)
        del ia, ib, ba, bb
    want = list_hash_torch(full[keep])
    return {"workload": "the headline scene: clear+extend+par_sort+par_scan_filtered with the box-overlap predicate evaluated on the device "
                        "inside the scan (bp_layer_scan_filtered_aabb_device), steady state, %d steps" % steps,
            "ms_per_step": ms, "pairs": n_out, "candidate_pairs_unfiltered": int(full.shape[0]),
            "value": n_out / (ms * 1e-3), "unit": "box-overlapping pairs/s", "list_hash": got,
            "parity": got == want,
            "parity_note": "list fingerprint == that of the unfiltered device list (parity_in_run) filtered with the same predicate by torch ops"}


def run_one_shot(bp, scenes, torch, np, dev, local_rank, n, seed, pairs_hint, entries_hint, repeats=3):
    """what a reference test does (tests/test_layer.rs:92-124): a FRESH layer, capacities given to the builder
    (with_index_capacity / with_collision_capacity), one extend + par_sort + par_scan.  No plan hints, no previous frame."""
    sysb, bounds, ids = scenes.gen_boxes(n, seed=seed)
    d_bounds = torch.from_numpy(bounds).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    times, n_out, st = [], 0, {}
    for _ in range(repeats):
        layer = (bp.LayerBuilder.new().with_min_depth(0).with_device(local_rank)
                 .with_index_capacity(entries_hint).with_collision_capacity(pairs_hint).build(bp.Index64_3D))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        layer.extend(sysb, d_bounds, d_ids)
        layer.par_sort()
        out = layer.scan_device()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        n_out, st = len(out), layer.stats()
        del out, layer
    return {"workload": "%d objects, a fresh layer per run built with with_index_capacity(%d).with_collision_capacity(%d): "
                        "extend + par_sort + par_scan once (%d runs)" % (n, entries_hint, pairs_hint, repeats),
            "ms_per_step": min(times), "ms_runs": [round(t, 4) for t in times], "pairs": n_out,
            "value": n_out / (min(times) * 1e-3), "unit": "pairs/s",
            "tree_sort_passes": st.get("tree_sort_passes"), "pair_sort_passes": st.get("pair_sort_passes"),
            "note": "ms_per_step = the fastest run (the first also pays one-time runtime initialisation of the kernels it is the first to launch)"}


def main():
    args = parse()
    import numpy as np
    import torch  # plumbing: device memory for the inputs, torch.distributed (RCCL) for N > 1
    import broadphase_rs_amd as bp
    from broadphase_rs_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    # BP_BENCH_DEVICE / BP_DIST_BACKEND=gloo: rehearse the multi-process path with every rank on ONE GPU (gloo stages
    # the exchanges through the host); the measured configuration is one GPU per rank over nccl = RCCL
    if os.environ.get("BP_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["BP_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded_path = world > 1 or args.sharded

    if not sharded_path:
        r = run_single(bp, scenes, torch, np, dev, local_rank, args.objects, args.seed, args.warmup, args.steps, args.aabb_filter)
        line = single_gpu_line(args, r, np)
        # in-run parity: the device list's fingerprint against the CPU row's (same scene, same seed)
        gpu_hash = list_hash_torch(r["out"].as_torch()) if r["n_pairs"] else [0, 0, 0]
        line["list_hash"] = gpu_hash
        line["set_hash"] = set_hash_torch(r["out"].as_torch()) if r["n_pairs"] else [0, 0, 0]
        del r["out"]
        if not args.no_1m and args.objects != 1_000_000 and not args.aabb_filter:
            r1 = run_single(bp, scenes, torch, np, dev, local_rank, 1_000_000, args.seed, 5, 50, detail=False, events=False)
            line["config_1m"] = {"workload": "configs[1]: 1M Index64_3D uniform-random AABBs, single MI355X, clear+extend+par_sort+par_scan",
                                 "ms_per_step": r1["ms_per_step"], "pairs": r1["n_pairs"],
                                 "value": r1["n_pairs"] / (r1["ms_per_step"] * 1e-3), "unit": "pairs/s",
                                 "cold_frame_ms": r1["cold_frame_ms"], "host_syncs_per_step": r1["host_syncs_per_step"],
                                 "tree_sort_passes": r1["stats"]["tree_sort_passes"], "pair_sort_passes": r1["stats"]["pair_sort_passes"]}
            del r1
        if not args.no_extra and not args.aabb_filter:
            # sub-records the driver's default run verifies: configs[4]'s per-GPU share, the fused narrow phase, a one-shot layer
            line["aabb_filter"] = run_aabb_filter(bp, torch, np, dev, r, r["d_bounds"], 5)
            st10 = r["stats"]
            line["one_shot"] = run_one_shot(bp, scenes, torch, np, dev, local_rank, args.objects, args.seed,
                                            int(st10["pairs"] * 1.07), int(st10["entries"] * 1.07))
        r.pop("layer", None); r.pop("scene", None); r.pop("d_bounds", None)
        torch.cuda.empty_cache()
        if not args.no_extra and not args.aabb_filter:
            line["mixed_scale"] = run_mixed_scale(bp, scenes, torch, np, dev, local_rank, args.mixed_objects, args.seed, 5,
                                                  check=not args.no_cpu_baseline)
        if not args.no_moving and not args.aabb_filter:
            line["moving"] = run_moving(bp, scenes, torch, np, dev, local_rank, args.objects, args.seed, args.moving_frames,
                                        check_last=not args.no_cpu_baseline)
        if not args.no_cpu_baseline and not args.aabb_filter:
            cb = cpu_baseline(args.cpu_objects, args.seed)
            line["cpu_baseline"] = cb
            same_scene = args.cpu_objects == args.objects
            line["parity_in_run"] = (cb["list_hash"] == gpu_hash) if same_scene else None
            line["parity_note"] = ("fingerprint (count, sum p, sum p*(2i+1) mod 2^64, p = id<<32|id_) of the device pair list vs "
                                   "the CPU row's list on the same scene" if same_scene else
                                   "CPU sample is a different scene size: lists not comparable")
        else:
            line["cpu_baseline"] = None
            line["parity_in_run"] = None
        print(json.dumps(line))
        return

    # ---- N > 1 (or --sharded rehearsal): Morton-range shards over torch.distributed ---------------------------
    # librccl prints a version banner straight to file descriptor 1 when a communicator is created: keep stdout for the one
    # JSON line (everything until then goes to stderr)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    dist_backend = os.environ.get("BP_DIST_BACKEND", "nccl")
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(dist_backend)
    strong = args.scaling == "strong"
    n_global = args.objects if strong else args.objects * world
    n = n_global // world
    n_global = n * world
    # one global scene, sharded by object index: rank r owns ids [r*n, (r+1)*n); same density as
    # the reference's recipe, so the global side is cbrt(n_global / 0.001) + 5.5
    sysb, bounds, ids = scenes.gen_boxes_shard(n_global, rank, world, seed=args.seed) if world > 1 else \
        scenes.gen_boxes(n, seed=args.seed)
    d_bounds = torch.from_numpy(bounds).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    # Engine: the whole sharded pipeline behind the C ABI (include/bp_sharded.h: RCCL bound by the library itself --
    # what a Rust / C++ host calls), or the same steps driven from Python over torch.distributed (dist.py).  The C
    # engine is tried first; every rank must have joined its RCCL group for it to be used (agreed by an all-reduce).
    from broadphase_rs_amd import dist as bpd
    engine_name = os.environ.get("BP_SHARDED_ENGINE", "c")
    sharded = None
    if engine_name == "c" and dist_backend == "nccl":
        ok = 1
        try:
            from broadphase_rs_amd import sharded as bps
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid = torch.tensor(list(bps.Group.unique_id()), dtype=torch.uint8, device=dev)
            dist.broadcast(uid, 0)
            group = bps.Group.rccl(rank, world, bytes(uid.cpu().tolist()), local_rank)
            sharded = bps.CShardedLayer(group, rank, bp.Index64_3D, device=local_rank, id_count=n_global)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("rank %d: C sharded engine unavailable (%s): using torch.distributed\n" % (rank, e))
            ok = 0
        agree = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            sharded = None
        if sharded is not None:  # one untimed step through the engine: an error on any rank sends every rank to the other engine
            try:
                sharded.clear()
                sharded.extend(sysb, d_bounds, d_ids)
                sharded.par_scan_device(args.pair_order)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("rank %d: C sharded engine failed its first step (%s): using torch.distributed\n" % (rank, e))
                ok = 0
            agree = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 0:
                sharded = None
    if sharded is not None:
        engine_name = "c-abi (include/bp_sharded.h), RCCL bound by the library"

        def step(order):
            sharded.clear()
            sharded.extend(sysb, d_bounds, d_ids)
            return sharded.par_scan_device(order)
    else:
        engine_name = "python (broadphase_rs_amd/dist.py) over torch.distributed"
        layers = {}

        def step(order):
            if order not in layers:
                layers[order] = bpd.ShardedLayer(bp.Index64_3D, 0, device=local_rank, n_global=n_global, pair_order=order)
            lay = layers[order]
            lay.clear()
            lay.extend(sysb, d_bounds, d_ids)
            return lay.par_scan_device()

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def measure(order):
        """W warm-up + K timed steps of one pair order -> record (rank 0) with the job's pair count and fingerprints"""
        for _ in range(args.warmup):
            out = step(order)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(order)
        barrier()
        dt = time.perf_counter() - t0
        n_pairs_local = len(out)
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # fingerprints: the ordered one of the concatenated list (per-rank fingerprints combined with the ranks' list offsets,
        # rank order = list order) and the order-independent one of the union
        h = list_hash_torch(out.as_torch()) if n_pairs_local else [0, 0, 0]
        hs = set_hash_torch(out.as_torch()) if n_pairs_local else [0, 0, 0]
        hv = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in h + hs], dtype=torch.int64, device=dev)
        parts = [torch.empty_like(hv) for _ in range(world)]
        dist.all_gather(parts, hv)
        cnt = s1 = s2 = q2 = 0
        for pr in parts:
            k, a_, b_, _, _, c_ = [int(v) & M64 for v in pr.tolist()]
            s2 = (s2 + b_ + 2 * cnt * a_) & M64      # sum p (2 (i + cnt) + 1) = local + 2 cnt sum p
            s1 = (s1 + a_) & M64
            q2 = (q2 + c_) & M64
            cnt += k
        info = sharded.last if hasattr(sharded, "last") else layers[order].last
        route = info.get("pair_order", order) if isinstance(info, dict) else order
        # load balance over the ranks (entries of the shard, pairs its scan emitted, pairs it returns): max / mean
        mine = torch.tensor([float(info.get("entries_shard", 0)), float(info.get("raw_pairs", 0)), float(n_pairs_local)] if isinstance(info, dict)
                            else [0.0, 0.0, float(n_pairs_local)], dtype=torch.float64, device=dev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every)
        imbalance = {k: (float(every[:, j].max() / every[:, j].mean()) if float(every[:, j].mean()) > 0 else None)
                     for j, k in enumerate(("entries_shard", "raw_pairs", "pairs"))}
        return {"pair_order": route, "ms_per_step": dt / args.steps * 1e3, "pairs_total": cnt, "value": cnt / (dt / args.steps),
                "pairs_this_rank": n_pairs_local, "list_hash": [cnt, s1, s2] if route == "global" else None, "set_hash": [cnt, s1, q2],
                "stages": info, "imbalance_max_over_mean": imbalance}

    main_rec = measure(args.pair_order)
    other = "global" if args.pair_order == "local" else "local"
    other_rec = None if args.one_order else measure(other)
    if rank == 0:
        shard_stats = (sharded.shard_layer if hasattr(sharded, "shard_layer") else layers[args.pair_order].shard_layer).stats() \
            if (hasattr(sharded, "shard_layer") or layers) else {}
        order_note = {
            "local": "every rank returns the sorted duplicate-free pairs found in its own Morton range (boundary keys all-gathered, no pair "
                     "crosses the wire): the ranks' lists are disjoint and their union is the reference's set (set_hash = the single-GPU one)",
            "global": "raw pairs move to the owner of their first-id range: the ranks' lists, concatenated in rank order, ARE the reference's "
                      "sorted list (list_hash = the single-GPU one)"}
        line = {
            "metric": "candidate pairs/sec at 10M Index64_3D AABBs; achieved HBM GB/s vs peak",
            "value": main_rec["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_rec["ms_per_step"], "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("configs[3]: %d Index64_3D uniform-random AABBs in total over %d GPUs (gen_boxes recipe, sharded by "
                                    "object index), per-GPU extend, Morton-range exchange of entries + all-gather of boundary keys, "
                                    "shard sort + scan" % (n_global, world)) if strong else
                                   "%d Index64_3D uniform-random AABBs per GPU x %d GPUs (one %d-object scene)" % (n, world, n_global),
                       "objects_per_gpu": n, "objects_total": n_global, "entries_rank0_shard": shard_stats.get("entries"),
                       "pairs_rank0": main_rec["pairs_this_rank"], "pairs_total": main_rec["pairs_total"],
                       "pair_order": main_rec["pair_order"], "pair_order_note": order_note[main_rec["pair_order"]],
                       "parallelism": "morton-range shards x%d over RCCL" % world, "engine": engine_name},
            "roofline": None, "cpu_baseline": None,
            "objects_per_s": n_global / (main_rec["ms_per_step"] * 1e-3),
            "list_hash": main_rec["list_hash"], "set_hash": main_rec["set_hash"],
            "sharded": main_rec["stages"], "imbalance_max_over_mean": main_rec["imbalance_max_over_mean"],
        }
        if other_rec is not None:
            line["pair_order_" + other] = {"ms_per_step": other_rec["ms_per_step"], "value": other_rec["value"], "pairs_total": other_rec["pairs_total"],
                                           "pair_order": other_rec["pair_order"], "note": order_note[other_rec["pair_order"]],
                                           "list_hash": other_rec["list_hash"], "set_hash": other_rec["set_hash"], "sharded": other_rec["stages"]}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    dist.barrier()
    dist.destroy_process_group()


def single_gpu_line(args, r, np):
    n, st, prof, steps = args.objects, r["stats"], r["prof"], args.steps
    ms_per_step = r["ms_per_step"]
    n_pairs = r["n_pairs"]
    M, P_raw, P = st["entries"], st["raw_pairs"], st["pairs"]
    pk, pp = st["tree_sort_passes"], st["pair_sort_passes"]
    compact = bool(r.get("tree_compact_route"))
    table = kernel_table(n, M, P_raw, P, cells=bool(r.get("cell_route")), tree_cells=bool(r.get("tree_cell_route")), tree_compact=compact)
    k = r["detail_steps"]
    launches = {c: v[1] / k for c, v in prof.items()}
    per_step_ms = {c: v[0] / k for c, v in prof.items()}
    # SURVEY 8(d) bytes of each class per step = bytes per launch x launches (radix classes: one timed region per pass)
    survey_step_bytes = {c: table[c][1] * (launches[c] if c in ("tree_sort", "pair_sort") else 1.0) for c in table}
    kernels = {c: {"kernel": table[c][0], "ms_per_step": per_step_ms[c], "launches_per_step": launches[c],
                   "survey_bytes_per_launch": table[c][1],
                   "GBps": (survey_step_bytes[c] / (per_step_ms[c] * 1e-3) / 1e9) if per_step_ms[c] > 0 else 0.0,
                   "frac_of_peak": (survey_step_bytes[c] / (per_step_ms[c] * 1e-3) / 1e9 / HBM_PEAK_GBS) if per_step_ms[c] > 0 else 0.0}
               for c in table}
    for c in prof:
        if c not in kernels:
            kernels[c] = {"ms_per_step": per_step_ms[c], "launches_per_step": launches[c]}
    # dominant kernel: largest total time per step over ALL classes; its duration comes from HIP events on the layer's
    # stream inside the timed region
    dom_cls, dom_prof = r["dom_cls"], r["dom_prof"]
    dom_name, bytes_per_launch = table[dom_cls]
    per_launch_ms = dom_prof[0] / dom_prof[1] if dom_prof[1] else 0.0
    achieved = bytes_per_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
    # bytes this pipeline really has to move per step (its own kernels' reads + writes)
    moved = {
        # (compact records: one 8-byte word per entry from the emit pass to the finish instead of a 12-byte (index, id) entry)
        "encode": 28.0 * n + 4.0 * n + 16.0 * n * 2 + (8.0 if compact else 12.0) * M,  # bounds + ids read once, the 16-byte object record written and read, entries written
        "tree_sort": (16.0 if compact else 24.0) * M * pk,
        "tree_finish": (8.0 if compact else 12.0) * M + (18.0 if compact else 21.0) * M,  # record read; key, {id, ctx} word and (compact route, round 5) the 2-byte depth / range-length word written (+ id, box byte off the compact route, where they are derived from the words on demand)
        # compact route (round 5): {id, ctx} words + depth / range-length words read -- no key; else keys + {id, ctx} words; packed pairs written
        "scan_emit": (10.0 if compact else 16.0) * M + 8.0 * P_raw,
        "pair_sort": 16.0 * P_raw * pp,
        "dedup": 8.0 * P_raw + 8.0 * P,
    }
    moved_total = sum(moved.values())
    ss_classes = ("tree_hist", "tree_sort", "tree_finish", "scan_count", "scan_emit")
    ss_ms = sum(per_step_ms[c] for c in ss_classes)
    ss_moved = moved["tree_sort"] + moved["tree_finish"] + moved["scan_emit"]
    P_lit = st.get("raw_candidates", 0) or P_raw
    survey = {"B_encode": 28.0 * n + 12.0 * M, "B_treesort": (8.0 + 24.0 * 8) * M, "B_scan": 12.0 * M + 8.0 * P_lit,
              "B_pairsort": (8.0 + 16.0 * 8) * P_lit, "B_dedup": 8.0 * P_lit + 8.0 * P}
    line = {
        "metric": "candidate pairs/sec at 10M Index64_3D AABBs; achieved HBM GB/s vs peak" if not args.aabb_filter else
                  "VARIANT (not the headline metric): box-overlapping pairs/sec with the device AABB narrow phase fused into the scan",
        "value": n_pairs / (ms_per_step * 1e-3), "unit": "pairs/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,  # the N = 1 point of either curve
        "dtype": "u64", "data": "synthetic",
        "pair_sort_route": "cells (two claiming passes; every timed frame)" if r.get("cell_route") else "claiming pass + stable one-sweep pass",
        "tree_sort_route": ("cells (two claiming passes; every timed frame)" if r.get("tree_cell_route") else "claiming pass + stable one-sweep pass") +
                           ("; compact 8-byte records (every timed frame)" if compact else ""),
        "config": {"workload": ("%s%dM Index64_3D uniform-random AABBs, single MI355X (gen_boxes recipe, density 0.001, "
                                "sizes U[1,10)), clear+extend+par_sort+par_scan, inputs and pair list HBM-resident"
                                % ("configs[2]: " if n == 10_000_000 else "", n // 1_000_000)) if n % 1_000_000 == 0
                               else "%d Index64_3D AABBs, single MI355X" % n,
                   "objects": n, "entries": M, "raw_pairs": P_raw, "raw_candidates_examined": P_lit, "pairs": P,
                   "tree_sort_passes": pk, "pair_sort_passes": pp, "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bytes_per_launch,
                     "avg_launch_ms": per_launch_ms,
                     "launches_per_step": dom_prof[1] * max(1, r.get("event_every", 1)) / steps if dom_prof[1] else 0.0,
                     "timed_launches": dom_prof[1],
                     "event_sampling": "HIP events on the layer's stream inside the timed region, around every launch of the class in every "
                                       "%d-th frame (an event pair costs ~5 us of stream bubble)" % max(1, r.get("event_every", 1)),
                     "chosen_by": "largest total time per step over all kernel classes (warm-up pass with every class timed); a radix class is "
                                  "one launch per pass (two different kernels): achieved = bytes per pass / the passes' mean duration",
                     "bytes_model": "SURVEY.md 8(d) per-unit figures x this run's N / M / P_raw / P"},
        "cold_frame_ms": r["cold_frame_ms"], "first_frame_ms": r["first_frame_ms"], "cold_plan": r["cold_plan"],
        "cold_note": "cold_frame_ms: buffers allocated, plan hints forgotten (the plan of a layer's first frame); first_frame_ms: a "
                     "fresh layer, every hipMalloc included; ms_per_step: steady state (hints from the previous frame)",
        "host_syncs_per_step": r["host_syncs_per_step"],
        "step_moved": {"bytes": moved_total, "GBps": moved_total / (ms_per_step * 1e-3) / 1e9,
                       "frac_of_peak": moved_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "note": "reads + writes this pipeline's own kernels must make per step, over the whole step's wall time"},
        # north_star's target is stated on sort+scan (tree sort through the scan, SURVEY 8(d) B_sort+scan); the headline
        # fraction prices the bytes these kernels really move; the literal 8-digit model is kept as a secondary figure
        "sort_scan": {"ms_per_step": ss_ms, "moved_bytes": ss_moved,
                      "frac_of_peak": ss_moved / max(1e-12, ss_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "survey_model_bytes": survey["B_treesort"] + survey["B_scan"],
                      "survey_model_frac": (survey["B_treesort"] + survey["B_scan"]) / max(1e-12, ss_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "frac_of_peak = bytes moved by tree-sort passes + finish + scan / their kernel time / 8 TB/s; "
                              "survey_model_frac prices a literal 8-digit LSD sort (bytes this pipeline does not move: 2 + LDS finish "
                              "instead of 8 passes) and is NOT a bandwidth figure"},
        "survey_model": dict(survey, B_total=sum(survey.values()), note="SURVEY.md 8(d) byte model of a literal 8-digit pipeline on the "
                             "same N / M / P (for reference only; see step_moved for the bytes really moved)"),
        "objects_per_s": n / (ms_per_step * 1e-3),
        "kernels": kernels,
        "kernels_note": "per-class times from %d extra steps after the timed region with every class bracketed by HIP events (event "
                        "pairs cost stream bubbles: these steps run slower than ms_per_step); roofline.avg_launch_ms comes from events "
                        "inside the timed region" % k,
    }
    # HBM traffic per launch of the dominant kernel from the rocprofv3 PMC passes of the same command
    # (profiles/traffic.json, written by profiles/summarize.py with the guide's gfx950 corrections); null when absent
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        # the class may run several kernels per step ("a + b (...)"): the launches' traffic averages like their durations do
        names = [t.strip() for t in dom_name.split(" (")[0].split(" + ")]
        hit = []
        for nm in names:
            key = nm.split("<")[0]
            want_noval = "NoVal" in nm
            exact = ", " in nm   # a full instantiation name (as rocprof prints it): radix_claim_kernel<NoVal, true>
            hit += [v for kk, v in tj.get("bytes_per_launch", {}).items()
                    if (kk.startswith(nm) if exact else (kk.startswith(key) and (("NoVal" in kk) == want_noval or "radix" not in key)))]
        if hit and tj.get("objects") == n:
            line["roofline"]["traffic"] = sum(hit) / len(hit)   # the two instantiations of a radix pass (first pass / later passes) average
            line["roofline"]["traffic_source"] = tj.get("source")
    except (OSError, ValueError):
        pass
    return line


if __name__ == "__main__":
    main()
