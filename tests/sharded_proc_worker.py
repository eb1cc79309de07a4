"""One RANK of tests/test_gpu_sharded_procs.py, as its own process: the C sharded engine (include/bp_sharded.h) over its RCCL
binding, with BP_RCCL_LIBRARY pointing at the shared-memory stand-in (tests/cpp/fake_rccl.cpp).  Every rank uses GPU 0.
   python tests/sharded_proc_worker.py <rank> <world> <uid hex> <out dir> <scene> <n total> <pair order> <frames> [fail_rank [id|die]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    rank, world, uid, out_dir, scene, n_total, order, frames = (int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3]), sys.argv[4],
                                                                 sys.argv[5], int(sys.argv[6]), sys.argv[7], int(sys.argv[8]))
    fail_rank = int(sys.argv[9]) if len(sys.argv) > 9 else -1
    fail_mode = sys.argv[10] if len(sys.argv) > 10 else "id"   # "id": an id beyond id_count; "die": the process leaves before the last frame's first collective
    import broadphase_rs_amd as bp
    from broadphase_rs_amd import scenes, sharded
    group = sharded.Group.rccl(rank, world, uid, 0)
    sh = sharded.CShardedLayer(group, rank, bp.Index64_3D, device=0, id_count=n_total)
    status = "ok"
    try:
        for frame in range(frames):
            if scene == "powerlaw":
                sysb, b, ids = scenes.gen_boxes_powerlaw_shard(n_total, rank, world, seed=21 + frame, alpha=2.5)
            else:
                sysb, b, ids = scenes.gen_boxes_shard(n_total, rank, world, seed=3 + frame)
                if scene == "giants":   # three coarse objects per rank: boundary keys for every peer
                    L = float(sysb[1][0])
                    b[0] = [0, 0, 0, L, L, L]
                    b[1] = [0.1 * L * (rank + 1) / world, 0, 0, 0.1 * L * (rank + 1) / world + 0.4 * L, 0.5 * L, 0.45 * L]
                    b[2] = [0.3 * L, 0.3 * L, 0.3 * L, 0.62 * L, 0.62 * L, 0.62 * L]
            if rank == fail_rank and frame == frames - 1 and fail_mode == "die":
                open(os.path.join(out_dir, "status_%d.txt" % rank), "w").write("left")
                os._exit(0)              # no goodbye: its peers find out from their bounded waits
            if rank == fail_rank and frame == frames - 1:
                ids = ids.copy()
                ids[5] = n_total + 7     # an id beyond id_count: refused -- on EVERY rank, through the gathered status words
            sh.clear()
            sh.extend(sysb, b, ids)
            out = sh.par_scan(order)
            np.save(os.path.join(out_dir, "in_%d_%d.npy" % (frame, rank)), np.concatenate([b, ids.astype(np.float64).reshape(-1, 1)], axis=1))
            np.save(os.path.join(out_dir, "sys_%d.npy" % frame), np.asarray(sysb, dtype=np.float32))
            np.save(os.path.join(out_dir, "out_%d_%d.npy" % (frame, rank)), out)
        info = sh.last
        open(os.path.join(out_dir, "info_%d.txt" % rank), "w").write(repr(info))
    except bp.BroadphaseError as e:
        status = "error: %s" % (e,)
    open(os.path.join(out_dir, "status_%d.txt" % rank), "w").write(status)
    del sh
    group.destroy()


if __name__ == "__main__":
    main()
