"""N > 1 path on CPU: world_size-2 gloo run of the band split + all-gather that
bench.py uses on RCCL.  The band renderer here is the CPU oracle (test infrastructure)
-- what is under test is the sharding / reassembly logic, not the kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from svo_raytracer_amd.tiles import band_rows, gather_bands, gather_bands_to_root, stripe_layout, deinterleave


def test_band_rows_cover_frame_exactly_once():
    for h in (1080, 2160, 96, 100, 7, 8, 9):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(h, dtype=np.int32)
            rpr = None
            for r in range(world):
                y0, y1, rows = band_rows(h, world, r)
                assert y0 % 8 == 0 and rows % 8 == 0
                assert rpr in (None, rows)
                rpr = rows
                assert y0 == min(r * rows, ((h + 7) // 8) * 8)
                seen[y0:y1] += 1
            assert (seen == 1).all(), (h, world)
            assert rpr * world >= h


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_path, to_root):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(64)
    y0, y1, rpr = band_rows(h, world, rank)
    band = oracle.render(pool, w, h, CAMERAS["K1"], 2, 0, rows=(y0, y1), want_hits=False)
    color = torch.zeros((rpr * world, w), dtype=torch.int32)
    depth = torch.zeros((rpr * world, w), dtype=torch.float32)
    color[y0:y1] = torch.from_numpy(band["rgba"].view(np.int32).reshape(h, w)[y0:y1].copy())
    depth[y0:y1] = torch.from_numpy(band["depth"][y0:y1].copy())
    if to_root:
        gather_bands_to_root(dist, color, rank, world, rpr)
        gather_bands_to_root(dist, depth, rank, world, rpr)
    else:
        gather_bands(dist, color, rank, rpr)
        gather_bands(dist, depth, rank, rpr)
    if rank == 0:
        np.savez(out_path, color=color.numpy()[:h], depth=depth.numpy()[:h])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("h,to_root", [(96, False), (100, False), (96, True), (100, True)])
def test_two_rank_band_split_reassembles_the_frame(tmp_path, h, to_root):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    w = 64
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), w, h, out, to_root), nprocs=2, join=True)
    z = np.load(out)
    pool, _ = scene.build_scene(64)
    full = oracle.render(pool, w, h, CAMERAS["K1"], 2, 0, want_hits=False)
    assert (z["color"].view(np.uint8).reshape(h, w, 4) == full["rgba"]).all()
    assert (z["depth"].view(np.uint32) == full["depth"].view(np.uint32)).all()


def test_stripe_layout_covers_every_tile_row_once():
    for h in (1080, 2160, 8640, 116, 7, 8, 9):
        for world in (1, 2, 3, 4, 8):
            tile_rows = (h + 7) // 8
            seen = np.zeros(tile_rows, dtype=np.int32)
            for r in range(world):
                first, step, n, out0, rpr = stripe_layout(h, world, r)
                assert out0 == r * rpr and n * 8 <= rpr
                for j in range(n):
                    seen[first + j * step] += 1
            assert (seen == 1).all(), (h, world)


def _stripe_worker(rank, world, port, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    pool, _ = scene.build_scene(64)
    first, step, n, out0, rpr = stripe_layout(h, world, rank)
    color = torch.zeros((rpr * world, w), dtype=torch.int32)
    for j in range(n):   # the band renderer here is the oracle, one 8-row stripe at a time
        y0 = (first + j * step) * 8
        y1 = min(h, y0 + 8)
        band = oracle.render(pool, w, h, CAMERAS["K1"], 2, 0, rows=(y0, y1), want_hits=False)
        color[out0 + j * 8:out0 + j * 8 + (y1 - y0)] = torch.from_numpy(
            band["rgba"].view(np.int32).reshape(h, w)[y0:y1].copy())
    gather_bands_to_root(dist, color, rank, world, rpr)
    if rank == 0:
        np.save(out_path, deinterleave(color, world, rpr, h).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_interleaved_stripes_reassemble_the_frame(tmp_path):
    import svo_raytracer_amd.scene as scene
    from svo_raytracer_amd.cameras import CAMERAS
    from oracle import oracle
    w, h = 64, 100
    out = str(tmp_path / "stripes.npy")
    mp.spawn(_stripe_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    got = np.load(out)
    pool, _ = scene.build_scene(64)
    full = oracle.render(pool, w, h, CAMERAS["K1"], 2, 0, want_hits=False)
    assert (got.view(np.uint8).reshape(h, w, 4) == full["rgba"]).all()
