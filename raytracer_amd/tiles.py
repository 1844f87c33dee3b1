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


# ---- packed tile exchange (collective #2) --------------------------------------------------------
# A rank's tiles of one frame travel as one contiguous buffer: its owned tiles in row-major tile
# order, each tile x tile pixels row-major inside the tile, RGBA8; pixels of an edge tile that lie
# outside the frame are zero padding.  Device side: rt_hip_pack_tiles_device / rt_hip_unpack_tiles_device.

def owned_tiles(width: int, height: int, tile: int, rank: int, nranks: int) -> np.ndarray:
    """int32 ids (ty * tiles_x + tx) of the tiles `rank` owns, in row-major tile order."""
    tiles_x, tiles_y = -(-width // tile), -(-height // tile)
    ty, tx = np.divmod(np.arange(tiles_x * tiles_y), tiles_x)
    return np.nonzero((tx + ty) % nranks == rank)[0].astype(np.int32)


def max_owned_tiles(width: int, height: int, tile: int, nranks: int) -> int:
    return max(len(owned_tiles(width, height, tile, r, nranks)) for r in range(nranks))


def _tile_pixels(width, height, tile, tile_list):
    """(ntiles * tile * tile,) pixel index of every packed slot, -1 for padding."""
    tiles_x = -(-width // tile)
    ty, tx = np.divmod(np.asarray(tile_list, dtype=np.int64), tiles_x)
    ly, lx = np.divmod(np.arange(tile * tile), tile)
    x = tx[:, None] * tile + lx[None, :]
    y = ty[:, None] * tile + ly[None, :]
    pix = y * width + x
    pix[(x >= width) | (y >= height)] = -1
    return pix.reshape(-1)


def pack_tiles(frame_rgba: np.ndarray, width: int, height: int, tile: int, tile_list) -> np.ndarray:
    """frame_rgba: (height*width, 4) uint8 -> (len(tile_list) * tile * tile, 4) uint8."""
    pix = _tile_pixels(width, height, tile, tile_list)
    out = np.zeros((pix.shape[0], 4), np.uint8)
    ok = pix >= 0
    out[ok] = frame_rgba.reshape(-1, 4)[pix[ok]]
    return out


def unpack_tiles(packed: np.ndarray, width: int, height: int, tile: int, tile_list, frame_rgba: np.ndarray) -> None:
    """Scatter a packed buffer back into frame_rgba ((height*width, 4) uint8), in place."""
    pix = _tile_pixels(width, height, tile, tile_list)
    ok = pix >= 0
    frame_rgba.reshape(-1, 4)[pix[ok]] = packed.reshape(-1, 4)[: pix.shape[0]][ok]
