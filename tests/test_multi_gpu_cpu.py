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


# ---- the strip plan of the sharded semi-dense flow (SURVEY 8e bullet 2; csrc/sdof.hip flow_impl with a communicator) ----
def _strip_worker(rank, world, port, nrows, ncols, patch, nscales, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vpp_amd import multi_gpu as mg
    rng = np.random.default_rng(3)
    kps = np.stack([rng.integers(0, nrows, 4000), rng.integers(0, ncols, 4000)], 1)
    res = {}
    for s, fr in enumerate(mg.flow_map_rows(nrows, patch, nscales)):
        fc = mg.flow_map_rows(ncols, patch, nscales)[s]
        per, mem, bounds = mg.strip_plan(fr, world)
        lo, hi = bounds[rank]
        # the claim of :114-143 restricted to this rank's rows: lowest keypoint index per cell (sdof_claim_kernel's atomicMin)
        owner = torch.full((mem, fc), 2**31 - 1, dtype=torch.int32)
        pf = (kps // (1 << s)) // patch
        for i in range(len(kps) - 1, -1, -1):
            r, c = int(pf[i, 0]), int(pf[i, 1])
            if lo <= r < hi and c < fc:
                owner[r, c] = i
        mg.all_gather_rows_inplace(owner, per, rank, world)
        res[f"owner{s}"] = owner[:fr].numpy()
    np.savez(os.path.join(out_dir, f"strip_rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nrows", [(2, 2160), (3, 1080), (2, 243)])
def test_flow_strip_plan_and_row_gather_gloo(tmp_path, world, nrows):
    """Every flow-map row of every scale belongs to exactly one rank, the chunks are equal-sized, and the in-place all-gather of the ranks'
    rows rebuilds on every rank the claim map a single process computes."""
    from vpp_amd import multi_gpu as mg
    ncols, patch, nscales = 320, 5, 3
    for fr in mg.flow_map_rows(nrows, patch, nscales):
        per, mem, bounds = mg.strip_plan(fr, world)
        assert mem >= fr and mem - fr < world and bounds[0][0] == 0 and bounds[-1][1] == fr
        assert all(bounds[g][1] == bounds[g + 1][0] for g in range(world - 1)) and all(hi - lo <= per for lo, hi in bounds)
    assert mg.flow_map_rows(2160, 5, 3) == [432, 217, 109]
    port = 29500 + (os.getpid() % 2000) + 11 * world + nrows % 97
    mp.spawn(_strip_worker, args=(world, port, nrows, ncols, patch, nscales, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(3)
    kps = np.stack([rng.integers(0, nrows, 4000), rng.integers(0, ncols, 4000)], 1)
    got = [np.load(os.path.join(str(tmp_path), f"strip_rank{r}.npz")) for r in range(world)]
    for s, fr in enumerate(mg.flow_map_rows(nrows, patch, nscales)):
        fc = mg.flow_map_rows(ncols, patch, nscales)[s]
        want = np.full((fr, fc), 2**31 - 1, np.int32)
        pf = (kps // (1 << s)) // patch
        ok = (pf[:, 0] < fr) & (pf[:, 1] < fc)
        np.minimum.at(want, (pf[ok, 0], pf[ok, 1]), np.nonzero(ok)[0].astype(np.int32))
        for r in range(world):
            np.testing.assert_array_equal(got[r][f"owner{s}"], want)
