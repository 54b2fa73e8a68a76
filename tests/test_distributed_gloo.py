"""World-size-2 (and 3) gloo runs of the band sharding + gather used for multi-GPU rendering (CPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpuraytracer_amd.distributed import band_for_rank, gather_bands
    full = torch.arange(height * width * 4, dtype=torch.float32).reshape(height, width, 4)
    first, rows = band_for_rank(height, world, rank)
    band = full[first:first + rows].clone()           # stands for the band this rank rendered
    img = gather_bands(band, height, width)
    dist.barrier()
    if rank == 0:
        ok = img is not None and torch.equal(img, full)
        np.save(result_path, np.array([1 if ok else 0]))
    else:
        assert img is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 36), (2, 37), (3, 10)])
def test_gather_bands_over_gloo(tmp_path, world, height):
    path = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, 16, path), nprocs=world, join=True)
    assert np.load(path)[0] == 1
