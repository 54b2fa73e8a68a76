"""Image-row sharding across ranks (one process per GPU) and the per-frame gather.

Rays are independent, so the image rows are split over the ranks, all rays of a pixel on one rank; the
scene and its hierarchy are replicated (each rank builds locally). Rows are either dealt out in turn
(rank r owns rows r, r + world, ...: neighbouring rows cost about the same, so every rank gets the same
load) or cut into contiguous bands. The only exchange on the path is the gather of the finished rows to
rank 0, one collective per displayed frame (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def band_for_rank(height, world_size, rank):
    """(first_row, num_rows) of a rank; rows are dealt as evenly as possible, earlier ranks get the extras."""
    base, extra = divmod(int(height), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def rows_for_rank(height, world_size, rank):
    """(first_row, num_rows, row_stride) of a rank when the rows are dealt out in turn."""
    height, world_size, rank = int(height), int(world_size), int(rank)
    return rank, max(0, (height - rank + world_size - 1) // world_size), world_size


def gather_rows(rows, height, width, channels=4, dst=0):
    """Gathers every rank's [num_rows, width, channels] float32 rows (dealt out in turn, rows_for_rank) to rank
    `dst`; returns the assembled [height, width, channels] image there and None elsewhere. Ranks may differ by
    one row, so each sends the row count of rank 0."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return rows
    rows_max = rows_for_rank(height, world, 0)[1]
    if dist.get_backend() == "gloo" and rows.is_cuda:      # gloo has no device gather: stage through the host
        rows = rows.cpu()
    padded = rows
    if rows.shape[0] != rows_max:
        padded = torch.zeros((rows_max, width, channels), dtype=rows.dtype, device=rows.device)
        padded[:rows.shape[0]] = rows
    padded = padded.contiguous()
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, out, dst=dst)
    if rank != dst:
        return None
    image = torch.empty((height, width, channels), dtype=rows.dtype, device=rows.device)
    for r in range(world):
        image[r::world] = out[r][:rows_for_rank(height, world, r)[1]]
    return image


def gather_bands(band, height, width, channels=4, dst=0):
    """Gathers every rank's [rows, width, channels] float32 band to rank `dst`; returns the full
    [height, width, channels] image there and None elsewhere. Bands may differ by one row, so each is
    padded to the largest band for the collective."""
    world, rank = dist.get_world_size(), dist.get_rank()
    rows_max = band_for_rank(height, world, 0)[1]
    padded = band
    if band.shape[0] != rows_max:
        padded = torch.zeros((rows_max, width, channels), dtype=band.dtype, device=band.device)
        padded[:band.shape[0]] = band
    padded = padded.contiguous()
    if world == 1:
        return band
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, out, dst=dst)
    if rank != dst:
        return None
    parts = [out[r][:band_for_rank(height, world, r)[1]] for r in range(world)]
    return torch.cat(parts, dim=0)
