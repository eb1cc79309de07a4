#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Layer extend -> sort -> scan path on MI355X.

Metric (BASELINE.json): candidate pairs/sec at 10M Index64_3D AABBs; achieved HBM GB/s vs peak.
A "step" is one pass of the hot path over one batch of synthetic AABBs that are already resident
in HBM: clear + extend + par_sort + par_scan, result (sorted, duplicate-free pair list) left in HBM.
value = unique pairs produced by all ranks / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--objects M]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`
(one rank per GPU, RCCL); scaling is weak: every rank contributes `--objects` objects of one
global scene, entries and pairs are exchanged by Morton / id range (broadphase_rs_amd/dist.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--objects", type=int, default=10_000_000, help="objects per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-objects", type=int, default=10_000_000, help="size of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="run the multi-GPU pipeline even with one rank (rehearsal)")
    ap.add_argument("--pair-order", choices=["global", "local"], default="global",
                    help="VARIANT: 'local' leaves every rank's pairs where they were found (set-equal, not the sorted partition)")
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


def cpu_baseline(n_objects, seed):
    """The oracle (C restatement of the reference's algorithm and structure) timed on this host.
    kind = "port": the Rust crate itself cannot be built here (no rustc/cargo).  Bounded sample:
    the same gen_boxes recipe and density at n_objects objects, full extend+par_sort+par_scan."""
    import oracle  # checker only: never on the measured GPU path
    from broadphase_rs_amd import scenes
    sysb, bounds, ids = scenes.gen_boxes(n_objects, seed=seed)
    threads = usable_cpus(oracle.lib().orc_hw_threads())
    rows = {}
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
        del lay, pairs
    best = max(rows.values(), key=lambda r: r["pairs_per_s"])
    return {"value": best["pairs_per_s"] / 1e6, "unit": "Mpairs/s", "cores": threads, "kind": "port",
            "sample": "gen_boxes(%d objects, density 0.001, seed %d): clear+extend+par_sort+par_scan, OpenMP "
                      "C restatement with the reference's structure; rows: %s" % (
                          n_objects, seed, json.dumps({k: round(v["pairs_per_s"] / 1e6, 3) for k, v in rows.items()})),
            "rows": rows}


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
    dist_backend = os.environ.get("BP_DIST_BACKEND", "nccl")
    dist = None
    sharded_path = world > 1 or args.sharded
    if sharded_path:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(dist_backend)

    n = args.objects
    n_global = n * world
    # one global scene, sharded by object index: rank r owns ids [r*n, (r+1)*n); same density as
    # the reference's recipe, so the global side is cbrt(n_global / 0.001) + 5.5
    sysb, bounds, ids = scenes.gen_boxes_shard(n_global, rank, world, seed=args.seed) if world > 1 else \
        scenes.gen_boxes(n, seed=args.seed)
    d_bounds = torch.from_numpy(bounds).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
    torch.cuda.synchronize()

    if not sharded_path:
        layer = bp.LayerBuilder.new().with_min_depth(0).with_device(local_rank).build(bp.Index64_3D)

        def step():
            layer.clear()
            layer.extend(sysb, d_bounds, d_ids)
            layer.par_sort()
            if args.aabb_filter:
                return layer.scan_filtered_aabb_device(d_bounds)
            return layer.scan_device()
        engine = layer
    else:
        from broadphase_rs_amd import dist as bpd
        sharded = bpd.ShardedLayer(bp.Index64_3D, 0, device=local_rank, n_global=n_global, pair_order=args.pair_order)

        def step():
            sharded.clear()
            sharded.extend(sysb, d_bounds, d_ids)
            return sharded.par_scan_device()
        engine = sharded.shard_layer

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up (untimed) with every kernel class timed: finds the dominant kernel.  Each HIP event pair
    # costs a few microseconds of stream bubble (all classes: ~0.2 ms per step), so the TIMED region
    # carries events only around the dominant kernel -- the one `roofline` reports.
    DOM_CAND = {"scan_emit": 1.0, "dedup": 1.0, "pair_sort": None}   # class -> launches per step (None: measured)
    engine.profile_reset()
    engine.profile_select(None)
    engine.profile_enable(True)
    for _ in range(args.warmup):
        out = step()
    engine.profile_enable(False)
    wprof = engine.profile()
    per_launch = {c: (wprof[c][0] / max(wprof[c][1], 1)) for c in DOM_CAND}
    dom_cls = max(per_launch, key=per_launch.get) if args.warmup > 0 and any(per_launch.values()) else "scan_emit"
    engine.profile_reset()
    engine.profile_select([dom_cls])
    engine.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    engine.profile_enable(False)
    dom_prof = engine.profile()[dom_cls]          # (ms, launches) of the dominant class inside the timed region
    # per-kernel breakdown: extra steps after the timed region, every class timed
    detail_steps = max(1, min(args.steps, 3))
    engine.profile_reset()
    engine.profile_select(None)
    engine.profile_enable(True)
    for _ in range(detail_steps):
        out = step()
    torch.cuda.synchronize()
    engine.profile_enable(False)
    n_pairs_local = len(out)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([n_pairs_local], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_pairs = int(c.item())
    else:
        n_pairs = n_pairs_local

    if rank == 0:
        st = engine.stats()
        prof = engine.profile()
        ms_per_step = dt / args.steps * 1e3
        value = n_pairs / (dt / args.steps)
        M, P_raw, P = st["entries"], st["raw_pairs"], st["pairs"]
        pk, pp = st["tree_sort_passes"], st["pair_sort_passes"]
        # ALGORITHMIC bytes per step and per kernel class (SURVEY.md 8(d) per-unit figures x measured
        # N / M / P_raw / P).  P_raw is what the scan actually emitted: with
        # duplicate suppression at the source it equals P.
        model = {
            "encode": 64.0 * n + 13.0 * M,                    # bounds+ids read twice, counts w+r, entries + box written
            "tree_hist": 0.0,                                 # digits are counted inside encode_emit
            "tree_sort": 24.0 * M * pk + 25.0 * M,            # key+id per pass (r+w) + finish (12 r + 13 w)
            "scan_count": 21.0 * M,                           # entry_ctx: key, id, box read; {id, context} word written
            "scan_emit": 16.0 * M + 8.0 * P_raw,              # scan_expand: keys + {id, context} words read; packed pairs written
            "pair_hist": 0.0,                                 # digits are counted inside the scan
            "pair_sort": 16.0 * P_raw * pp,
            "dedup": 16.0 * P_raw,                            # finish / unique: pairs read, ABI layout written
            "misc": 0.0,
        }
        launches = {k: v[1] / detail_steps for k, v in prof.items()}
        per_step_ms = {k: v[0] / detail_steps for k, v in prof.items()}
        kernels = {k: {"ms_per_step": per_step_ms[k], "timed_regions_per_step": launches[k], "model_bytes_per_step": model[k],
                       "GBps": (model[k] / (per_step_ms[k] * 1e-3) / 1e9) if per_step_ms[k] > 0 else 0.0}
                   for k in prof}
        # dominant kernel = the kernel (by name) with the largest time per step.  Radix passes are
        # timed per launch, so a class total is launches x one kernel; the scan is a single launch.
        cand = {
            "scan_expand_kernel (overlap scan, single pass)": ("scan_emit", 1.0, model["scan_emit"]),
            "segment_finish_kernel (pair segments -> final list)": ("dedup", 1.0, model["dedup"]),
            "radix_onesweep_kernel<u64,NoVal> (pair-sort pass)": ("pair_sort", max(launches["pair_sort"], 1.0), 16.0 * P_raw),
        }
        dom_name = [k for k in cand if cand[k][0] == dom_cls][0]
        _, _, bytes_per_launch = cand[dom_name]
        dom_launches = dom_prof[1] / args.steps if dom_prof[1] else 1.0
        per_launch_ms = dom_prof[0] / dom_prof[1] if dom_prof[1] else 0.0   # HIP events inside the timed region
        achieved = bytes_per_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        # the radix pass is the HBM-bound kernel of the path: always reported as well
        rp_ms = per_step_ms["pair_sort"] / max(launches["pair_sort"], 1.0)
        rp_achieved = 16.0 * P_raw / (rp_ms * 1e-3) / 1e9 if rp_ms > 0 else 0.0
        b_total = sum(model.values())
        # SURVEY.md 8(d)'s own byte model (what a literal 8-digit LSD pipeline moves: p_k = p_p = 8 passes,
        # the reference's raw pair stream before dedup), evaluated on this run's N / M / P_raw / P
        P_lit = st.get("raw_candidates", 0) or P_raw
        survey = {"B_encode": 28.0 * n + 12.0 * M, "B_treesort": (8.0 + 24.0 * 8) * M, "B_scan": 12.0 * M + 8.0 * P_lit,
                  "B_pairsort": (8.0 + 16.0 * 8) * P_lit, "B_dedup": 8.0 * P_lit + 8.0 * P}
        survey_total = sum(survey.values())
        line = {
            "metric": "candidate pairs/sec at 10M Index64_3D AABBs; achieved HBM GB/s vs peak" if not (args.aabb_filter or args.pair_order != "global") else
                      "VARIANT (not the headline metric): pairs left on the rank that found them (set-equal, unsorted across ranks)" if not args.aabb_filter else
                      "VARIANT (not the headline metric): box-overlapping pairs/sec with the device AABB narrow phase fused into the scan",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%dM Index64_3D uniform-random AABBs per GPU (gen_boxes recipe, density 0.001, "
                                   "sizes U[1,10)), clear+extend+par_sort+par_scan, inputs and pair list HBM-resident"
                                   % (n // 1_000_000) if n % 1_000_000 == 0 else "%d Index64_3D AABBs per GPU" % n,
                       "objects_per_gpu": n, "objects_total": n_global, "entries": M, "raw_pairs": P_raw,
                       "pairs_rank0": P, "pairs_total": n_pairs, "tree_sort_passes": pk, "pair_sort_passes": pp,
                       "parallelism": "1 GPU" if world == 1 else "morton-range shards x%d, RCCL all-to-all" % world},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": per_launch_ms, "launches_per_step": dom_launches},
            "roofline_radix_pass": {"bound": "hbm", "kernel": "radix_onesweep_kernel<u64,NoVal> (pair-sort pass)",
                                    "achieved": rp_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": rp_achieved / HBM_PEAK_GBS, "bytes_per_launch": 16.0 * P_raw, "avg_launch_ms": rp_ms},
            "step_model": {"bytes": b_total, "GBps": b_total / (ms_per_step * 1e-3) / 1e9,
                           "frac_of_peak": b_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "bytes the kernels of THIS pipeline must move (digit skipping, partial sorts, duplicates "
                                   "suppressed at the source already taken out)"},
            "survey_model": dict(survey, raw_pairs_reference=P_lit, B_total=survey_total,
                                 ms_at_peak=survey_total / (HBM_PEAK_GBS * 1e9) * 1e3,
                                 frac_of_peak=survey_total / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 note="SURVEY.md 8(d) byte model of a literal 8-digit LSD pipeline on the same N/M/P; the step "
                                      "beats it by moving fewer bytes, not by exceeding HBM bandwidth (see step_model)"),
            # SURVEY 8(d) also asks for objects/s and for the sort+scan part alone (tree sort through dedup)
            "objects_per_s": n_global / (dt / args.steps),
            "sort_scan": {"ms_per_step": sum(per_step_ms[k] for k in ("tree_hist", "tree_sort", "scan_count", "scan_emit", "pair_hist",
                                                                      "pair_sort", "dedup")),
                          "pairs_per_s": n_pairs / max(1e-12, 1e-3 * sum(per_step_ms[k] for k in (
                              "tree_hist", "tree_sort", "scan_count", "scan_emit", "pair_hist", "pair_sort", "dedup"))),
                          # north_star's target: >= 50 % of the HBM roofline on sort+scan, where SURVEY 8(d) defines
                          # B_sort+scan = B_treesort + B_scan (tree sort and scan proper, before the pair sort)
                          "tree_sort_and_scan_ms": sum(per_step_ms[k] for k in ("tree_hist", "tree_sort", "scan_count", "scan_emit")),
                          "survey_B_sort_scan": survey["B_treesort"] + survey["B_scan"],
                          "survey_frac_of_peak": (survey["B_treesort"] + survey["B_scan"]) / max(1e-12, 1e-3 * sum(
                              per_step_ms[k] for k in ("tree_hist", "tree_sort", "scan_count", "scan_emit"))) / 1e9 / HBM_PEAK_GBS,
                          "note": "kernel time of tree sort + scan + pair sort + dedup in the per-class steps (see kernels_note); "
                                  "survey_frac_of_peak prices SURVEY's model bytes, which this pipeline does not all move "
                                  "(see step_model for the bytes it does)"},
            "kernels": kernels,
            "kernel_ms_per_step": per_step_ms,
            "kernels_note": "per-class times from %d extra steps after the timed region with every class bracketed by HIP "
                            "events (those steps run ~0.2 ms slower than ms_per_step); roofline.avg_launch_ms comes from "
                            "events inside the timed region" % detail_steps,
        }
        # HBM traffic per launch of the dominant kernel from the rocprofv3 PMC passes of the same
        # command (profiles/traffic.json, written by profiles/summarize.py: 2 x FETCH_SIZE + WRITE_SIZE
        # KiB, the guide's gfx950 correction); null when no such profile has been committed
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            key = {"scan_emit": "scan_expand_kernel", "dedup": "segment_finish_kernel"}.get(dom_cls, "radix_onesweep_kernel<u64, NoVal")
            hit = [v for k, v in tj.get("bytes_per_launch", {}).items() if k.startswith(key)]
            if hit and tj.get("objects") == n:
                line["roofline"]["traffic"] = hit[0]
                line["roofline"]["traffic_source"] = tj.get("source")
        except (OSError, ValueError):
            pass
        if sharded_path:
            line["sharded"] = sharded.last
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_objects, args.seed)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
