"""N>1 path on CPU: two gloo ranks shard the keypoints, each tracks its slice (with the CPU oracle standing in for the
device kernel — this test is about the sharding and the exchange, not the kernel), one all-gather rebuilds the set, and the
result equals the single-process run bit for bit."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _track(orc, kps):
    import pyr
    from test_oracle_algos import lk_scene
    from util import u8_image
    from vpp_amd import image as vi
    f1, f2, _ = lk_scene(160, 200, 10)
    i1, i2 = u8_image(f1), u8_image(f2)
    hp1, hp2 = pyr.host_pyramid(orc, i1, 3, 5), pyr.host_pyramid(orc, i2, 3, 5)
    hg = pyr.host_grad_pyramid(orc, hp1[0], 3, 5, vi.F32)
    k = kps.copy()
    orc.orc_pyrlk_match(vi.desc_array(hp1), vi.desc_array(hg), vi.desc_array(hp2), 3, k.ctypes.data_as(ctypes.c_void_p), len(k), 7,
                        ctypes.c_float(1e-4), ctypes.c_float(500.0), 30, ctypes.c_float(0.01), 0, None)
    return k


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyr
    from oracle import binding
    from vpp_amd import multi_gpu as mg
    orc = binding.load()
    kps = pyr.make_keypoints(pyr.grid_keypoints(160, 200, n, margin=24))
    lo, hi = mg.shard_bounds(n, rank, world)
    mine = _track(orc, kps[lo:hi])
    shard = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
    full = mg.all_gather_records(shard, n, rank, world)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(150, 2), (151, 2), (151, 3)])  # even and ragged shards, 2 and 3 ranks
def test_keypoint_sharding_and_all_gather_gloo(orc, tmp_path, n, world):
    import pyr
    from vpp_amd import multi_gpu as mg
    port = 29500 + (os.getpid() % 2000) + n + 7 * world
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    want = _track(orc, pyr.make_keypoints(pyr.grid_keypoints(160, 200, n, margin=24))).view(np.uint8).reshape(-1)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        np.testing.assert_array_equal(got, want)
    assert [mg.shard_bounds(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]


def test_cpp_harness_shard_plan(tmp_path):
    """benchmarks/shard_plan.hh — the slice / pad / unpad arithmetic of benchmarks/pyrlk_shard_bench.cc — for world sizes 1-8, even and ragged."""
    import subprocess
    exe = str(tmp_path / "shard_plan_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "shard_plan_test.cc"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "shard_plan_test ok" in out.stdout, out.stderr
