"""Row-band partition of an image over ranks — the host-side twin of `pixel_id()` /
`local_rows64()` / `pt_bands_kernel` in csrc/, used by bench.py and the multi-rank tests.

Rows are cut into bands of `band_rows` rows; band b belongs to rank b % world.  Work-item ids stay
global (the reference's `id = row*W + col`, src/cl/main.cl:5-9), so a pixel's RNG stream and running
mean do not depend on the partition: ranks write disjoint rows of zero-initialised frames and a sum
(RCCL reduce on GPUs, any reduce on the CPU) reassembles the single-GPU image bit for bit.
"""
import numpy as np


def rank_of_row(row, world, band_rows):
    return (row // band_rows) % world


def rows_of_rank(height, rank, world, band_rows):
    """Sorted array of the image rows `rank` renders."""
    rows = np.arange(height)
    return rows[(rows // band_rows) % world == rank]


def local_pixel_count(width, height, rank, world, band_rows):
    return int(len(rows_of_rank(height, rank, world, band_rows))) * width


def id_ranges_of_rank(width, height, rank, world, band_rows):
    """Contiguous [begin, end) work-item id ranges of `rank`, one per band."""
    out = []
    nb = (height + band_rows - 1) // band_rows
    for b in range(rank, nb, world):
        lo, hi = b * band_rows, min((b + 1) * band_rows, height)
        out.append((lo * width, hi * width))
    return out


def packed_ids_of_rank(width, height, rank, world, band_rows):
    """Global work-item id of local pixel t = 0, 1, ... of `rank` — `pixel_id()` of csrc/pt_render_kernels.h, which is
    also the order in which pt_combine_bands packs a rank's bands into ONE message (its bands, top to bottom)."""
    r = id_ranges_of_rank(width, height, rank, world, band_rows)
    return np.concatenate([np.arange(b, e, dtype=np.int64) for b, e in r]) if r else np.zeros(0, np.int64)


def pack_bands(frame, rank, world, band_rows):
    """[H, W, C] frame -> [local pixels, C]: this rank's bands, compact (what travels to the root)."""
    h, w = frame.shape[:2]
    return frame.reshape(h * w, -1)[packed_ids_of_rank(w, h, rank, world, band_rows)]


def unpack_bands(frame, packed, rank, world, band_rows):
    """The root's half: scatter a peer's message into the full frame, in place."""
    h, w = frame.shape[:2]
    frame.reshape(h * w, -1)[packed_ids_of_rank(w, h, rank, world, band_rows)] = packed
