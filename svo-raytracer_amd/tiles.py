"""Screen-tile sharding of one frame over the GPUs of a node.

The unit is the reference's 8x8 work-group tile (svotrace.comp:648); a rank renders a
contiguous band of tile rows, all bands padded to the same number of pixel rows so
that one all-gather of equal chunks reassembles the frame."""

TILE = 8


def band_rows(height, world, rank):
    """Rows [y0, y1) rendered by `rank`, and the (padded) rows every rank's band occupies."""
    tile_rows = (height + TILE - 1) // TILE
    per = (tile_rows + world - 1) // world
    rows_per_rank = per * TILE
    y0 = min(rank * rows_per_rank, ((height + TILE - 1) // TILE) * TILE)
    y1 = min(height, y0 + rows_per_rank)
    y1 = max(y1, y0)
    return y0, y1, rows_per_rank


def gather_bands(dist, full, rank, rows_per_rank):
    """All-gather the equal-sized bands in place: `full` is [rows_per_rank * world, W, ...],
    this rank's band already sits at its final position inside it."""
    mine = full[rank * rows_per_rank:(rank + 1) * rows_per_rank]
    dist.all_gather_into_tensor(full, mine)
    return full


def gather_bands_to_root(dist, full, rank, world, rows_per_rank, dst=0, force=False, scratch=None):
    """Gather the equal-sized bands to rank `dst` only (the north-star exchange step: one
    frame owner, every peer sends its band over its direct xGMI link).  `full` is
    [rows_per_rank * world, W, ...] on every rank; only dst's copy ends up complete."""
    mine = full[rank * rows_per_rank:(rank + 1) * rows_per_rank]
    if world == 1 and not force:
        return full
    if rank == dst:
        parts = [full[r * rows_per_rank:(r + 1) * rows_per_rank] for r in range(world)]
        # the root's own band is already in place; give gather a scratch slot for it
        parts[dst] = scratch if scratch is not None else mine.clone()
        dist.gather(mine, gather_list=parts, dst=dst)
    else:
        dist.gather(mine, gather_list=None, dst=dst)
    return full


def stripe_layout(height, world, rank):
    """Interleaved sharding for load balance: rank r renders tile rows r, r + world, r + 2*world, ...
    Returns (first_tile_row, tile_row_step, n_tile_rows, out_row0, rows_per_rank): the arguments of
    svo_set_stripes plus the padded band size every rank's packed stripes occupy in the gather buffer."""
    tile_rows = (height + TILE - 1) // TILE
    per = (tile_rows + world - 1) // world
    n = max(0, (tile_rows - rank + world - 1) // world)
    rows_per_rank = per * TILE
    return rank, world, n, rank * rows_per_rank, rows_per_rank


def deinterleave(full, world, rows_per_rank, height):
    """[world * rows_per_rank, W, ...] stripe-major (as gathered) -> [height, W, ...] in frame order."""
    per = rows_per_rank // TILE
    shp = full.shape
    v = full.reshape(world, per, TILE, *shp[1:])
    perm = (1, 0, 2) + tuple(range(3, v.dim() if hasattr(v, "dim") else v.ndim))
    v = v.permute(*perm) if hasattr(v, "permute") else v.transpose(perm)
    return v.reshape(per * world * TILE, *shp[1:])[:height]
