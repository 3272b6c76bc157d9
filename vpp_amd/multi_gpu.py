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


# ---- row strips of the flow maps: the sharded semi-dense flow (vpp_semi_dense_optical_flow_sharded, csrc/sdof.hip flow_impl) ----
def flow_map_rows(nrows, patchsize, nscales):
    """Rows of the flow map at every scale: pf_domain = nrows / patchsize (semi_dense_optical_flow.hpp:68-69), then 1 + n / 2 per pyramid
    level (pyramid.hh:154)."""
    fr = [nrows // patchsize]
    for _ in range(1, nscales):
        fr.append(1 + fr[-1] // 2)
    return fr


def strip_plan(fr, world):
    """(per, rows_of_memory, [(lo, hi) per rank]) for a map of fr rows: rank g owns the rows [g per, (g + 1) per) clipped to fr, per =
    ceil(fr / world); the map is allocated with per * world rows so that every rank's rows are one chunk of per * pitch bytes at
    rank * chunk and ONE in-place all-gather completes the map on every rank.  (Strips are per scale: a coarse cell is read from the
    complete, gathered coarser map, so the strips of different scales need not nest.)"""
    per = -(-fr // world)
    return per, per * world, [(min(fr, g * per), min(fr, (g + 1) * per)) for g in range(world)]


def all_gather_rows_inplace(rows, per, rank, world):
    """rows: 2-D tensor (per * world, pitch) of which this rank filled rows [rank per, (rank + 1) per).  In-place all-gather."""
    if world == 1:
        return rows
    assert rows.shape[0] == per * world and rows.is_contiguous()
    if dist.get_backend() == "gloo" and rows.is_cuda:
        host = rows.cpu(); all_gather_rows_inplace(host, per, rank, world); rows.copy_(host); return rows
    mine = rows[rank * per:(rank + 1) * per].clone()   # RCCL gathers in place (send = recv + rank * chunk); gloo wants disjoint buffers
    dist.all_gather_into_tensor(rows.view(-1), mine.view(-1))
    return rows
