"""The RCCL paths with MORE THAN ONE RANK on a one-GPU box: both C++ harnesses (one process per rank, C ABI only) are started twice
on device 0.  RCCL refuses two ranks on one GPU ("Duplicate GPU detected") when it believes they share a host, so each rank is given
its own NCCL_HOSTID: the ranks then look like two nodes and talk through RCCL's socket transport over the loopback interface — slow, but
every collective of the product path really executes: ncclAllGather of the keypoint records (vpp_allgather_tracks), the in-place
all-gathers of frame rows and flow-map rows (vpp_allgather_rows, vpp_semi_dense_optical_flow_sharded) and the grouped ncclSend / ncclRecv
of the halo rows (vpp_halo_exchange).  Each harness checks its result against a single-rank run of the same call and reports mismatches."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(exe, world, args, timeout=300):
    assert os.path.exists(exe), f"{exe} was not built (python -c 'import __graft_entry__ as g; g.build()')"
    uid = os.path.join(tempfile.mkdtemp(prefix="vpp_uid_"), "id")
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"NCCL_HOSTID": f"vpp-one-gpu-rank-{r}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1",
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0", "VPP_SHARD_EAGER": "1"})   # eager: a socket-transport collective is host-driven, keep it out of a graph
        procs.append(subprocess.Popen([exe, str(r), str(world), uid] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, o, e in outs:
        assert rc == 0, f"rank exited with {rc}\nstdout: {o[-2000:]}\nstderr: {e[-3000:]}"
    line = [ln for ln in outs[0][1].strip().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world", [2, 3, 8])   # 8: the node's rank count — 1 250 keypoints per rank of configs[3]'s 10 000
def test_pyrlk_keypoint_shards_and_rccl_all_gather_two_ranks_one_gpu(world):
    res = run_ranks(os.path.join(ROOT, "benchmarks", "pyrlk_shard_bench"), world, [3, 10000])
    assert res["mismatched_vs_single_rank"] == 0, res
    assert res["keypoints_per_rank"] == -(-10000 // world)


@pytest.mark.parametrize("world,pairs", [(8, 8), (3, 4)])
def test_pyrlk_frame_pair_batches_sharded_over_ranks_one_gpu(world, pairs):
    """Round 6: the keypoint-sharded step over F frame pairs — every rank matches its slice of all F pairs in ONE vpp_pyrlk_match_batch launch, ONE RCCL all-gather
    carries the F x slice records; per pair the gathered records equal a single-rank vpp_pyrlk_match over all 10 000 keypoints (8 ranks x 8 pairs x 1 250: the
    shape whose rate the bench line projects)."""
    res = run_ranks(os.path.join(ROOT, "benchmarks", "pyrlk_shard_bench"), world, [2, 10000, pairs], timeout=600)
    assert res["mismatched_vs_single_rank"] == 0, res
    assert res["frame_pairs_per_step"] == pairs and res["keypoints_per_rank"] == -(-10000 // world)


@pytest.mark.parametrize("world,shape", [(2, (480, 640)), (3, (540, 960)), (2, (2160, 3840)), (8, (480, 640)), (8, (1120, 1280))])   # 8 ranks (the node), strips of 60 and 140 rows
def test_flow_strips_halo_exchange_and_map_gathers_two_ranks_one_gpu(world, shape):
    """Row exchange + sharded semi-dense flow + halo exchange + FAST-9 on strips: identical to the single-rank calls (BASELINE configs[4] at 4K)."""
    res = run_ranks(os.path.join(ROOT, "benchmarks", "flow_strip_bench"), world, [2, shape[0], shape[1]], timeout=600)
    assert res["mismatched_rows_or_flow_records"] == 0, res
    assert res["mismatched_fast9_strips"] == 0, res
