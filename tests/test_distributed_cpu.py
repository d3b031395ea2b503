"""world_size-2 (and 3) gloo tests of the exchange step and of the sharded multi-view fusion logic (no GPU:
the HIP registration is replaced by an injected stub; the exchange code path is the one the GPUs run)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_views, q):
    try:
        import importlib
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        r3d = importlib.import_module("3d_reconstruction_project_amd")
        D = r3d.distributed
        # --- all_gather_v with ragged (and empty) blocks
        mine = np.full((rank * 3, 4), float(rank)) + np.arange(rank * 3)[:, None]
        blocks = D.all_gather_v(mine)
        assert len(blocks) == world
        for r, b in enumerate(blocks):
            assert b.shape == (r * 3, 4) and (b[:, 0] == r + np.arange(r * 3)).all()
        # --- the exchange step on tensors: ragged views, several per rank, blocks come back as contiguous [n, 3] planes
        import torch
        owned0 = D.shard_views(n_views, rank, world)
        loc = {v: torch.arange(2 * (5 + v) * 3, dtype=torch.float64).reshape(2, 5 + v, 3) + 1000.0 * v for v in owned0}
        allv = D.gather_views(loc, n_views)
        assert sorted(allv) == list(range(n_views))
        for v in range(n_views):
            want = torch.arange(2 * (5 + v) * 3, dtype=torch.float64).reshape(2, 5 + v, 3) + 1000.0 * v
            assert torch.equal(allv[v], want) and allv[v][0].is_contiguous() and allv[v][1].is_contiguous()
        Ts = D.gather_transforms({v: np.eye(4) * (v + 1) for v in owned0}, n_views)
        assert all(np.array_equal(Ts[v], np.eye(4) * (v + 1)) for v in range(n_views))
        try:
            D.gather_views({}, n_views) if owned0 else None           # a rank must bring exactly its own views
            assert not owned0
        except ValueError:
            pass
        try:                                                           # host tensors without stubs: no CPU fallback
            r3d.pipeline.multi_view_fuse_tensors(loc, n_views)
            raise AssertionError("expected the product path to refuse host tensors")
        except RuntimeError as e:
            assert "no CPU fallback" in str(e)
        # --- sharded multi-view fusion: view v = the same patch displaced by a known translation t_v
        owned = D.shard_views(n_views, rank, world)
        rng = np.random.default_rng(0)
        base = np.concatenate([rng.random((500, 2)), np.zeros((500, 1))], 1)
        shifts = {v: np.array([0.01 * v, -0.02 * v, 0.005 * v]) for v in range(n_views)}
        clouds = {v: r3d.PointCloud(base[: 400 + 10 * v] - shifts[v], normals=np.tile([0, 0, 1.0], (400 + 10 * v, 1)))
                  for v in owned}

        def stub_register(src, tgt):                       # exact for pure translations of the same leading points
            T = np.eye(4)
            T[:3, 3] = tgt[:400, :3].mean(0) - src[:400, :3].mean(0)
            return T
        fused, Ts = r3d.pipeline.multi_view_fuse(clouds, n_views, register=stub_register)
        assert sorted(Ts) == list(range(n_views))
        for v in range(n_views):
            assert np.abs(Ts[v][:3, 3] - shifts[v]).max() < 1e-12
        assert len(fused.points) == sum(400 + 10 * v for v in range(n_views)) and fused.has_normals()
        assert np.abs(fused.points[:400] - base[:400]).max() < 1e-12      # every view lands back on the base patch
        q.put((rank, "ok", float(fused.points.sum())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc() + str(e)))


@pytest.mark.parametrize("world,n_views", [(2, 8), (3, 5)])
def test_exchange_and_sharded_fusion_gloo(world, n_views):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    assert len({r[2] for r in res}) == 1            # every rank holds the same fused cloud


def test_world_size_one_needs_no_process_group(r3d):
    import torch
    blocks = r3d.distributed.all_gather_v(np.ones((3, 2)))
    assert len(blocks) == 1 and blocks[0].shape == (3, 2)
    if not torch.cuda.is_available():                       # single rank, host tensors: the exchange is the identity
        loc = {v: torch.full((2, 3 + v, 3), float(v), dtype=torch.float64) for v in range(3)}
        out = r3d.distributed.gather_views(loc, 3)
        assert all(torch.equal(out[v], loc[v]) for v in range(3))
        assert np.array_equal(r3d.distributed.gather_transforms({0: np.eye(4)}, 1)[0], np.eye(4))
    assert r3d.distributed.shard_views(8, 1, 4) == [1, 5]
    assert sorted(sum((r3d.distributed.shard_views(8, r, 3) for r in range(3)), [])) == list(range(8))


def test_pointcloud_container_semantics(r3d):
    a = r3d.PointCloud(np.zeros((2, 3)), colors=np.ones((2, 3)))
    b = r3d.PointCloud(np.ones((3, 3)), colors=np.zeros((3, 3)), normals=np.tile([0, 0, 1.0], (3, 1)))
    a += b
    assert len(a.points) == 5 and a.has_colors() and not a.has_normals()     # legacy +=: attributes need both sides
    e = r3d.PointCloud()
    e += b
    assert e.has_normals() and e.has_colors() and len(e) == 3
