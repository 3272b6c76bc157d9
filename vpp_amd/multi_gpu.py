"""Keypoint sharding across the GPUs of one node: keypoints are independent (reference:
vpp/algorithms/pyrlk/pyrlk_match.hh:24-51), so rank g tracks the contiguous slice [g*N/G, (g+1)*N/G) against pyramids
every rank holds, and ONE collective — an all-gather of the fixed-size keypoint records — rebuilds the full set on every
rank (RCCL over xGMI on the GPU box: backend "nccl"; gloo in the CPU tests).  No reference counterpart: the reference is
single-process OpenMP."""
import torch
import torch.distributed as dist

RECORD_BYTES = 20  # keypoint<float>: position (2 x f32), velocity (2 x f32), age (i32) — vpp/core/keypoint_container.hh:13-25


def shard_bounds(n, rank, world):
    return rank * n // world, (rank + 1) * n // world


def all_gather_records(shard, n, rank, world, record_bytes=RECORD_BYTES):
    """shard: 1-D uint8 tensor holding this rank's records.  Returns the n*record_bytes tensor of all records, in index order."""
    if world == 1:
        return shard
    if dist.get_backend() == "gloo" and shard.is_cuda:  # CPU tests / 1-GPU smoke runs: gloo gathers host tensors
        return all_gather_records(shard.cpu(), n, rank, world, record_bytes).to(shard.device)
    sizes = [(shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0]) * record_bytes for r in range(world)]
    assert shard.numel() == sizes[rank]
    if len(set(sizes)) == 1:
        out = torch.empty(n * record_bytes, dtype=torch.uint8, device=shard.device)
        dist.all_gather_into_tensor(out, shard)
        return out
    # ragged shards: pad to the largest, one fixed-size all-gather, strip the padding
    mx = max(sizes)
    padded = torch.zeros(mx, dtype=torch.uint8, device=shard.device)
    padded[:shard.numel()] = shard
    out = torch.empty(world * mx, dtype=torch.uint8, device=shard.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)])
