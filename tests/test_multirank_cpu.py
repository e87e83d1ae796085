"""The N>1 path on the CPU: world_size-2 (and 3) `gloo` process groups run the same row-band
partition and the same combine the product runs over RCCL (pt_combine_bands / pt_combine_bands_rank:
every rank packs its bands into ONE message to the root, the root unpacks; nothing needs to be zero),
and the reduce(sum) of zero-filled frames that bench.py keeps as its fallback — with the oracle standing
in for the kernel (no GPU here).  What is checked is the sharding logic: disjoint cover, global ids,
the pack/unpack map, and that the combined frame is the single-rank image bit for bit."""
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


def test_pack_map_is_the_kernels_local_pixel_order():
    """packed_ids_of_rank restates pixel_id() (csrc/pt_render_kernels.h): local pixel t -> row band -> global id."""
    for (W, H, world, band) in ((33, 54, 3, 5), (7, 17, 2, 16), (5, 9, 4, 100), (3, 1, 8, 8), (16, 40, 8, 8)):
        total = 0
        for r in range(world):
            ids = partition.packed_ids_of_rank(W, H, r, world, band)
            lp = partition.local_pixel_count(W, H, r, world, band)
            assert len(ids) == lp
            for t in range(lp):      # pixel_id() as the kernel computes it
                lrow, col = divmod(t, W)
                lband, rin = divmod(lrow, band)
                row = (lband * world + r) * band + rin
                assert row < H and ids[t] == row * W + col
            total += lp
        assert total == W * H
    frame = np.arange(54 * 33 * 4, dtype=np.float32).reshape(54, 33, 4)
    out = np.full_like(frame, -1.0)
    for r in range(3):
        partition.unpack_bands(out, partition.pack_bands(frame, r, 3, 5), r, 3, 5)
    assert np.array_equal(out, frame)


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
    # the product's combine: ONE message per peer, nothing zeroed (the foreign rows are poisoned first)
    full = t.numpy().copy()
    full[other] = np.nan
    if rank != 0:
        dist.send(torch.from_numpy(np.ascontiguousarray(partition.pack_bands(full, rank, world, band))), dst=0)
    else:
        for r in range(1, world):
            msg = torch.empty((partition.local_pixel_count(W, H, r, world, band), 4), dtype=torch.float32)
            dist.recv(msg, src=r)
            partition.unpack_bands(full, msg.numpy(), r, world, band)
        np.save(out_path + ".packed.npy", full)
    # bench.py's fallback: reduce(sum) of zero-filled frames
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
    assert np.array_equal(np.load(out + ".packed.npy").view(np.uint32), want.view(np.uint32))
    assert want[..., :3].max() > 0
