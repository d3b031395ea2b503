"""Multi-GPU layer: one process per GPU, one view (stereo pair / frame) per rank, ONE exchange step.

The reference is single-process / single-GPU (every device is the literal "CUDA:0"); the only parallel axis this
build adds is data-parallel sharding of independent views.  SGM, reprojection, voxel
down-sampling and normals need no communication.  Registration-to-a-common-view needs every rank to see view 0's
cloud, and the mesher (mesh_reconstruction.py:22-37, rank 0) needs the fused cloud: both are served by a single
all-gather-v of the per-view clouds over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).
On the fully connected xGMI mesh an all-gather moves each shard once per peer link; no all-reduce or ring is needed.

The payload never leaves HBM under "nccl": clouds are torch tensors that wrap device memory the HIP library wrote
(r3d_disparity_to_cloud_resident), the collective runs on torch's current stream, which init() also makes the library's
stream, and the gathered blocks are handed back to the library as device pointers (r3d_icp_dev, r3d_transform_points_dev).
Only the row counts (a few int64 per rank) and the 4x4 transforms are read on the host.
"""
import os

import numpy as np


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_views(n_views, rank, world):
    """View v is owned by rank v % world (8 views on 8 GPUs: one each; fewer GPUs: round-robin)."""
    return [v for v in range(n_views) if v % world == rank]


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world_size():
    d = _dist()
    return d.get_world_size() if d is not None else 1


def init(backend=None, ctx=None):
    """Binds this rank to its GPU BEFORE any other GPU call and returns (rank, world, ctx): torch.cuda.set_device(LOCAL_RANK),
    a library Context on the same device whose stream is torch's current stream (so library kernels, RCCL collectives and
    torch ops are ordered without events), and the process group ("nccl" when a GPU is visible, else "gloo") if
    WORLD_SIZE > 1 (or R3D_FORCE_DIST is set, to rehearse the collective path with one rank).  Idempotent."""
    from . import _lib
    if _lib.torch_preloaded is False:
        raise RuntimeError("libr3d_hip.so was loaded into a torch-free process (R3D_NO_TORCH_PRELOAD=1): importing torch now would "
                           "bring a second HIP runtime that sees no device; start the process without that variable")
    import torch
    rank, local_rank, world = dist_env()
    have_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if have_gpu else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        if ctx is None:
            ctx = _lib.default_context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    import torch.distributed as dist
    if (world > 1 or os.environ.get("R3D_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, ctx


def exchange_device():
    """Where exchanged tensors must live: the current CUDA (HIP) device under "nccl", the host under "gloo" / no group."""
    import torch
    d = _dist()
    if d is not None and d.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    if d is None and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_gather_fixed(t):
    """All ranks pass a tensor of the SAME shape; returns a [world, *shape] tensor on the same device."""
    import torch
    d = _dist()
    # a one-rank group normally needs no collective; R3D_FORCE_DIST issues it anyway (rehearsal of the RCCL path on one GPU)
    if d is None or (d.get_world_size() == 1 and not os.environ.get("R3D_FORCE_DIST")):
        return t.unsqueeze(0)
    world = d.get_world_size()
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if d.get_backend() == "nccl":
        d.all_gather_into_tensor(out, t.contiguous())
    else:
        d.all_gather([out[r] for r in range(world)], t.contiguous())
    return out


def gather_views(local, n_views, planes=2):
    """The exchange step.  local: {view_id: tensor [planes, n_v, 3] float64} owned by this rank (plane 0 = points, plane 1 =
    normals), all on exchange_device().  Returns {view_id: tensor [planes, n_v, 3]} for ALL views on every rank; the tensors
    are views into one gathered buffer (each plane of each view is a contiguous [n_v, 3] block the library can read in place).
    Two collectives: a fixed-size header (view id, row count per owned view), then ONE all-gather of the payload padded to
    the largest rank.  Nothing but the header is read on the host."""
    import torch
    d = _dist()
    world = d.get_world_size() if d is not None else 1
    rank = d.get_rank() if d is not None else 0
    ids = sorted(local)
    if ids != shard_views(n_views, rank, world):
        raise ValueError(f"rank {rank} must own views {shard_views(n_views, rank, world)}, got {ids}")
    dev = exchange_device()
    for v in ids:
        t = local[v]
        if t.dim() != 3 or t.shape[0] != planes or t.shape[2] != 3 or t.dtype != torch.float64:
            raise ValueError(f"view {v}: expected a float64 [{planes}, n, 3] tensor, got {tuple(t.shape)} {t.dtype}")
        if t.device.type != dev.type:
            raise ValueError(f"view {v} lives on {t.device}, the exchange runs on {dev}")
    slots = (n_views + world - 1) // world
    hdr = torch.full((slots, 2), -1, dtype=torch.int64)
    for i, v in enumerate(ids):
        hdr[i, 0], hdr[i, 1] = v, local[v].shape[1]
    hdrs = _all_gather_fixed(hdr.to(dev)).cpu().numpy()                 # [world, slots, 2]: the only host read
    rows = np.where(hdrs[:, :, 0] >= 0, hdrs[:, :, 1], 0).sum(1)
    max_rows = max(int(rows.max()), 1)
    pad = torch.zeros((planes, max_rows, 3), dtype=torch.float64, device=dev)
    o = 0
    for v in ids:
        n = local[v].shape[1]
        pad[:, o:o + n] = local[v]
        o += n
    everyone = _all_gather_fixed(pad)                                    # [world, planes, max_rows, 3]
    out = {}
    for r in range(world):
        o = 0
        for v, n in hdrs[r]:
            if v < 0:
                continue
            out[int(v)] = everyone[r, :, o:o + int(n)]
            o += int(n)
    if sorted(out) != list(range(n_views)):
        raise RuntimeError(f"views missing after the exchange: have {sorted(out)}, want 0..{n_views - 1}")
    return out


def gather_transforms(mine, n_views):
    """mine: {view_id: 4x4 numpy} of the owned views -> {view_id: 4x4 numpy} of all views on every rank (one small collective)."""
    import torch
    d = _dist()
    world = d.get_world_size() if d is not None else 1
    slots = (n_views + world - 1) // world
    buf = torch.zeros((slots, 17), dtype=torch.float64)
    buf[:, 0] = -1
    for i, v in enumerate(sorted(mine)):
        buf[i, 0] = v
        buf[i, 1:] = torch.from_numpy(np.asarray(mine[v], dtype=np.float64).reshape(16))
    allb = _all_gather_fixed(buf.to(exchange_device())).cpu().numpy().reshape(-1, 17)
    out = {int(r[0]): r[1:].reshape(4, 4).copy() for r in allb if r[0] >= 0}
    if sorted(out) != list(range(n_views)):
        raise RuntimeError(f"transforms missing after the exchange: have {sorted(out)}")
    return out


def all_gather_v(rows):
    """All-gather of float64 row blocks [n_r, c] with different row counts (numpy in, list of numpy blocks out, rank order).
    Convenience for small host-side data (statistics, checksums); clouds go through gather_views, which keeps them in HBM."""
    import torch
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2:
        raise ValueError("all_gather_v expects a 2-D array")
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return [rows]
    dev = exchange_device()
    cnt = _all_gather_fixed(torch.tensor([rows.shape[0], rows.shape[1]], dtype=torch.int64, device=dev)).cpu().numpy()
    cols = {int(c[1]) for c in cnt if c[0] > 0} or {rows.shape[1]}
    if len(cols) != 1:
        raise ValueError(f"ranks disagree on the column count: {sorted(cols)}")
    ncol = cols.pop()
    pad = torch.zeros((max(int(cnt[:, 0].max()), 1), ncol), dtype=torch.float64, device=dev)
    if rows.shape[0]:
        pad[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    out = _all_gather_fixed(pad).cpu().numpy()
    return [out[r, :int(cnt[r, 0])] for r in range(len(cnt))]
