"""Image-band sharding across ranks (one process per GPU) and the per-frame tile gather.

Rays are independent, so the image is cut into contiguous bands of whole rows, one per rank; the
scene and its hierarchy are replicated (each rank builds locally). The only exchange on the path is
the gather of finished bands to rank 0, one collective per displayed frame (RCCL over xGMI on GPUs,
gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def band_for_rank(height, world_size, rank):
    """(first_row, num_rows) of a rank; rows are dealt as evenly as possible, earlier ranks get the extras."""
    base, extra = divmod(int(height), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


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
