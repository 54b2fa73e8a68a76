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


def _worker_rows(rank, world, port, height, width, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpuraytracer_amd.distributed import gather_rows, rows_for_rank
    full = torch.arange(height * width * 4, dtype=torch.float32).reshape(height, width, 4)
    first, rows, stride = rows_for_rank(height, world, rank)
    mine = full[first::stride].clone()                # stands for the rows this rank rendered
    assert mine.shape[0] == rows
    img = gather_rows(mine, height, width)
    dist.barrier()
    if rank == 0:
        ok = img is not None and torch.equal(img, full)
        np.save(result_path, np.array([1 if ok else 0]))
    else:
        assert img is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 36), (2, 37), (3, 10), (4, 2)])
def test_gather_rows_over_gloo(tmp_path, world, height):
    path = str(tmp_path / "ok.npy")
    mp.spawn(_worker_rows, args=(world, _free_port(), height, 16, path), nprocs=world, join=True)
    assert np.load(path)[0] == 1


def test_rows_for_rank_covers_every_row_once():
    from gpuraytracer_amd.distributed import rows_for_rank
    for height in (1, 2, 7, 36, 2160):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, n, stride = rows_for_rank(height, world, r)
                seen += list(range(first, height, stride))
                assert n == len(range(first, height, stride))
            assert sorted(seen) == list(range(height))


@pytest.mark.parametrize("world,height", [(2, 36), (2, 37), (3, 10)])
def test_gather_bands_over_gloo(tmp_path, world, height):
    path = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, 16, path), nprocs=world, join=True)
    assert np.load(path)[0] == 1
