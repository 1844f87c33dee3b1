"""Host mirror of the device tile-ownership rule (k_frame.hip: tile_is_mine): interleaved
tile x tile screen tiles dealt round robin along a row, rotated by the tile row.  Pure index
arithmetic: per-pixel shading does not depend on the owner, so the union over ranks is the
single-GPU frame bit for bit."""
import numpy as np


def tile_owner(width: int, height: int, tile: int, nranks: int) -> np.ndarray:
    """(height*width,) int32: rank that owns each texel (row-major, row 0 first)."""
    x = np.arange(width)[None, :] // tile
    y = np.arange(height)[:, None] // tile
    return ((x + y) % nranks).astype(np.int32).reshape(-1)


def photon_shard(total: int, rank: int, nranks: int):
    """[first, first+count) of the global photon index sequence owned by `rank`."""
    base, rem = divmod(total, nranks)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)
