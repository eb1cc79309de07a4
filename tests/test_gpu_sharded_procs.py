"""The C sharded engine (csrc/bp_sharded.hip, include/bp_sharded.h) with its ranks as PROCESSES: world 2 and 3 on one GPU, the
engine's RCCL code path (grouped sends / receives per peer pair, messages cut into rounds, the status words of the first
all-gather, the bounded waits) over a real process boundary -- through tests/cpp/fake_rccl.cpp, a stand-in for librccl at its C ABI
that moves the bytes through shared memory (BP_RCCL_LIBRARY).  The pool offers one GPU per box, so ncclSend / ncclRecv between
two physical GPUs stay unexercised; everything on this side of that ABI runs here.  Results against the oracle on the union scene."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "broadphase_rs_amd", "lib", "libfake_rccl.so")


def build_fake():
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(FAKE), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-x", "hip", src,
                               "-o", FAKE, "-lrt"])
    return FAKE


def run_world(tmp_path, world, scene, n_total, order, frames=2, fail_rank=-1, chunk_bytes=None, timeout=300, fail_mode="id", wait_ms=20000):
    env = dict(os.environ, BP_RCCL_LIBRARY=build_fake(), FAKE_RCCL_BOX_BYTES=str(24 << 20), BP_SHARDED_TIMEOUT_MS=str(wait_ms), FAKE_RCCL_TIMEOUT_MS=str(wait_ms))
    if chunk_bytes:
        env["BP_SHARDED_CHUNK_BYTES"] = str(chunk_bytes)
    # the unique id comes from the stand-in as well (any process may ask: no GPU work in it)
    uid = subprocess.check_output([sys.executable, "-c",
                                   "import sys; sys.path.insert(0, %r); from broadphase_rs_amd import sharded; print(sharded.Group.unique_id().hex())" % ROOT],
                                  env=env).decode().strip()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_proc_worker.py"), str(r), str(world), uid, str(tmp_path), scene,
                               str(n_total), order, str(frames)] + ([str(fail_rank), fail_mode] if fail_rank >= 0 else []), env=env)
             for r in range(world)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=timeout))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()          # (exact processes this test started)
            raise
    assert all(rc == 0 for rc in rcs), rcs
    return [open(os.path.join(str(tmp_path), "status_%d.txt" % r)).read() for r in range(world)]


def oracle_pairs(tmp_path, world, frame):
    import oracle
    sysb = np.load(os.path.join(str(tmp_path), "sys_%d.npy" % frame))
    ref = oracle.OracleLayer(2, 0)
    for r in range(world):
        a = np.load(os.path.join(str(tmp_path), "in_%d_%d.npy" % (frame, r)))
        ref.extend(sysb, np.ascontiguousarray(a[:, :6].astype(np.float32)), np.ascontiguousarray(a[:, 6].astype(np.uint32)))
    return ref.par_scan()


@pytest.mark.parametrize("world,scene,n_total,order,chunk", [(2, "giants", 60000, "global", None), (3, "giants", 60000, "local", None),
                                                             (3, "powerlaw", 45000, "global", 64 << 10), (2, "uniform", 50000, "local", 32 << 10)])
def test_ranks_as_processes_match_the_oracle(tmp_path, world, scene, n_total, order, chunk):
    status = run_world(tmp_path, world, scene, n_total, order, frames=2, chunk_bytes=chunk)
    assert status == ["ok"] * world, status
    for frame in range(2):
        expect = oracle_pairs(tmp_path, world, frame)
        outs = [np.load(os.path.join(str(tmp_path), "out_%d_%d.npy" % (frame, r))) for r in range(world)]
        got = np.concatenate(outs, axis=0)
        if order == "global":      # the ranks' lists concatenate to the reference's sorted list
            assert got.shape == expect.shape and np.array_equal(got, expect), (frame, got.shape, expect.shape)
        else:                      # disjoint sorted lists whose union is the reference's set
            p = (got[:, 0].astype(np.uint64) << np.uint64(32)) | got[:, 1].astype(np.uint64)
            e = (expect[:, 0].astype(np.uint64) << np.uint64(32)) | expect[:, 1].astype(np.uint64)
            assert len(p) == len(e) and np.array_equal(np.sort(p), e), (frame, len(p), len(e))
            for o in outs:
                q = (o[:, 0].astype(np.uint64) << np.uint64(32)) | o[:, 1].astype(np.uint64)
                assert np.all(q[1:] > q[:-1])


def test_an_id_beyond_id_count_on_one_process_is_refused_by_every_process(tmp_path):
    """the failure model across a process boundary: rank 1 is handed a bad id in its last frame; its status word travels in the
    step's first all-gather, every rank reads every rank's words, and all of them return BP_ERR_INVALID_ARG together -- nobody is
    left waiting in an exchange (the workers exit on their own, well inside the timeout)"""
    status = run_world(tmp_path, 3, "uniform", 45000, "global", frames=2, fail_rank=1, timeout=120)
    assert all(s.startswith("error") for s in status), status


def test_a_process_that_leaves_is_noticed_by_its_peers_within_the_bounded_wait(tmp_path):
    """rank 1 of 3 exits without a word before the second frame's first collective: the others must come back from that frame with an
    error (BP_ERR_COMM through the transport's error or the engine's own bounded wait) instead of waiting for ever -- and exit cleanly"""
    import time
    t0 = time.time()
    status = run_world(tmp_path, 3, "uniform", 45000, "local", frames=2, fail_rank=1, timeout=120, fail_mode="die", wait_ms=4000)
    assert status[1] == "left"
    assert status[0].startswith("error") and status[2].startswith("error"), status
    assert time.time() - t0 < 90
    # the first frame, which every rank took part in, is complete and correct
    exp = oracle_pairs(tmp_path, 3, 0)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out_0_%d.npy" % r)).reshape(-1, 2) for r in range(3)])
    got = got[np.lexsort((got[:, 1], got[:, 0]))]
    assert got.shape == exp.shape and np.array_equal(got, exp)
