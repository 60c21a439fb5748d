"""Image-level data parallelism over the GPUs of one node.

The reference is single-process / single-device (no distributed code at all).  Images are
independent units, so image i goes to rank i mod world_size, weights are replicated, and the only
exchange step is one gather of fixed-shape padded records
    {count int32 ; LAFs (N,2,3) float32 ; responses (N) float32 ; descriptors (N,128) float32}   = 4 + 540 N bytes / image
per batch (RCCL over xGMI when the backend is "nccl"; the same code runs on gloo in the CPU tests): either an all_gather
(every rank ends up with every record) or a gather to rank 0 (SURVEY.md section 8e allows either; 7/8 less xGMI traffic
on 8 GPUs).  A record travels as one row of a float32 tensor whose first 4 bytes ARE the int32 count (bit pattern, not a
float conversion).  No collective sits inside the per-image data path.
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Round-robin partition: item i -> rank i % world_size."""
    return list(range(rank, n_items, world_size))


def pack_records(results, n_cap, device):
    """List of per-image result dicts (capacity-sized tensors + `count`) -> one float32 tensor
    (n_img, 1 + n_cap * 135): [count, LAFs(6), resp(1), desc(128)] per row, zero padded."""
    rows = []
    for r in results:
        cnt = r["count"].to(torch.int32).view(1).view(torch.float32)          # int32 bit pattern in the first 4 bytes
        body = torch.cat([r["LAFs"].reshape(n_cap, 6), r["responses"].reshape(n_cap, 1), r["descriptors"].reshape(n_cap, 128)], dim=1)
        rows.append(torch.cat([cnt, body.reshape(-1)]))
    if not rows:
        return torch.zeros(0, 1 + n_cap * 135, dtype=torch.float32, device=device)
    return torch.stack(rows).to(device)


def pack_batched_records(results, n_cap):
    """Same record layout from the batched fused path: each result dict holds (B, n_cap, ...) tensors (or
    (n_cap, ...) for a single image) and `count` (B,).  Returns (sum of B, 1 + n_cap * 135), images in order."""
    rows = []
    for r in results:
        cnt = r["count"].to(torch.int32).contiguous().view(-1, 1).view(torch.float32)   # int32 bit pattern in the first 4 bytes
        b = cnt.size(0)
        body = torch.cat([r["LAFs"].reshape(b, n_cap, 6), r["responses"].reshape(b, n_cap, 1), r["descriptors"].reshape(b, n_cap, 128)],
                         dim=2)
        rows.append(torch.cat([cnt, body.reshape(b, n_cap * 135)], dim=1))
    return torch.cat(rows, dim=0)


def unpack_record(row, n_cap):
    n = int(row[:1].contiguous().view(torch.int32).item())
    body = row[1:].view(n_cap, 135)[:n]
    return {"LAFs": body[:, :6].reshape(n, 2, 3), "responses": body[:, 6], "descriptors": body[:, 7:]}


def record_counts(records):
    """(n, record) float32 rows -> (n,) int32 keypoint counts (the first 4 bytes of every record)."""
    return records[:, :1].contiguous().view(torch.int32).view(-1)


def gather_features_async(local_records, n_total, group=None, force=False, dst=None):
    """Starts the exchange and returns a `finish()` callable that waits for it and returns the records in global image
    order - lets the caller overlap the exchange of step k with the compute of step k+1 (bench.py).  dst=None: all_gather
    (every rank gets all records); dst=r: gather to GLOBAL rank r only (finish() returns None on the other ranks).  force=True runs
    the collective even in a 1-rank group (single-GPU dry run of the RCCL path)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return lambda: local_records
    world = dist.get_world_size(group)
    if dst is not None:
        return _gather_to_rank_async(local_records, n_total, group, dst, world)
    per_rank = -(-n_total // world)
    if local_records.size(0) < per_rank:
        pad = torch.zeros(per_rank - local_records.size(0), local_records.size(1), dtype=local_records.dtype, device=local_records.device)
        local_records = torch.cat([local_records, pad])
    dev = local_records.device
    if dist.get_backend(group) == "gloo" and local_records.is_cuda:
        local_records = local_records.cpu()
    local_records = local_records.contiguous()
    out = torch.empty((world * per_rank, local_records.size(1)), dtype=local_records.dtype, device=local_records.device)
    work = dist.all_gather_into_tensor(out, local_records, group=group, async_op=True)

    def finish():
        work.wait()
        keep = local_records  # noqa: F841  (input must stay alive until the collective has finished)
        return out.view(world, per_rank, -1).transpose(0, 1).reshape(per_rank * world, -1)[:n_total].to(dev)
    return finish


def _gather_to_rank_async(local_records, n_total, group, dst, world):
    per_rank = -(-n_total // world)
    if local_records.size(0) < per_rank:
        pad = torch.zeros(per_rank - local_records.size(0), local_records.size(1), dtype=local_records.dtype, device=local_records.device)
        local_records = torch.cat([local_records, pad])
    dev = local_records.device
    if dist.get_backend(group) == "gloo" and local_records.is_cuda:
        local_records = local_records.cpu()
    local_records = local_records.contiguous()
    me = dist.get_rank()                 # dst is a GLOBAL rank (what dist.gather expects), so compare with the global rank: with a sub-group
                                         # dist.get_rank(group) is the group-local rank and the wrong rank (or none) would allocate the output
    if group is not None and dst not in dist.get_process_group_ranks(group):
        raise ValueError("gather destination %d (global rank) is not a member of the group" % dst)
    parts = [torch.empty_like(local_records) for _ in range(world)] if me == dst else None
    work = dist.gather(local_records, parts, dst=dst, group=group, async_op=True)      # RCCL: grouped send / recv to one rank

    def finish():
        work.wait()
        keep = local_records  # noqa: F841
        if me != dst:
            return None
        return torch.stack(parts, 0).transpose(0, 1).reshape(per_rank * world, -1)[:n_total].to(dev)
    return finish


def gather_features(local_records, n_total, group=None):
    """all_gather of the padded records; returns (n_total, record) in global image order.
    Every rank must hold ceil-divided shards of equal length (pad with zero rows otherwise)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_records
    world = dist.get_world_size(group)
    per_rank = -(-n_total // world)
    if local_records.size(0) < per_rank:
        pad = torch.zeros(per_rank - local_records.size(0), local_records.size(1), dtype=local_records.dtype, device=local_records.device)
        local_records = torch.cat([local_records, pad])
    dev = local_records.device
    if dist.get_backend(group) == "gloo" and local_records.is_cuda:
        local_records = local_records.cpu()          # gloo has no CUDA all_gather (CPU tests / single-GPU dry runs of bench.py)
    out = torch.empty((world * per_rank, local_records.size(1)), dtype=local_records.dtype, device=local_records.device)
    dist.all_gather_into_tensor(out, local_records.contiguous(), group=group)      # one RCCL all-gather on backend "nccl"
    stacked = out.view(world, per_rank, -1).transpose(0, 1).reshape(per_rank * world, -1)   # row j*world + r = item j of rank r = global item j*world + r
    return stacked[:n_total].to(dev)
