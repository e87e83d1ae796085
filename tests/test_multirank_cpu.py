"""The N>1 path on the CPU: world_size-2 (and 3) `gloo` process groups run the same row-band
partition + reduce(sum) that bench.py runs over RCCL, with the oracle standing in for the kernel
(no GPU here).  What is checked is the sharding logic: disjoint cover, global ids, and that the
reduced frame is the single-rank image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from pathtracer_amd import partition, scenes
from pathtracer_amd import seeds as seedmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bands_cover_the_image_once():
    for (H, world, band) in ((1080, 8, 8), (54, 3, 5), (17, 2, 16), (9, 4, 100), (1, 8, 8), (2160, 7, 16)):
        seen = np.zeros(H, np.int32)
        for r in range(world):
            rows = partition.rows_of_rank(H, r, world, band)
            seen[rows] += 1
            assert partition.local_pixel_count(33, H, r, world, band) == 33 * len(rows)
            ids = partition.id_ranges_of_rank(33, H, r, world, band)
            assert sum(e - b for b, e in ids) == 33 * len(rows)
            assert all(partition.rank_of_row(b // 33, world, band) == r for b, _ in ids)
        assert (seen == 1).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, band, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = oracle_lib.load_oracle()
    sc = scenes.main_lua_frame(5)
    s = seedmod.glibc_rand_stream(3)
    frame = oracle_lib.aligned_zeros(W * H * 4)
    for b, e in partition.id_ranges_of_rank(W, H, rank, world, band):
        oracle.render(sc, W, H, 50, s, image=frame, id_begin=b, id_end=e, nthreads=2)
    t = torch.from_numpy(frame.reshape(H, W, 4).copy())
    mine = partition.rows_of_rank(H, rank, world, band)
    other = np.setdiff1d(np.arange(H), mine)
    assert not t.numpy()[other].any()          # a rank touches only its own rows
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,band", [(2, 8), (3, 5)])
def test_gloo_band_partition_reduces_to_the_single_rank_image(tmp_path, oracle, world, band):
    import torch.multiprocessing as mp
    W, H = 48, 27
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, band, out), nprocs=world, join=True)
    got = np.load(out)
    want = oracle.render(scenes.main_lua_frame(5), W, H, 50, seedmod.glibc_rand_stream(3))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert want[..., :3].max() > 0
