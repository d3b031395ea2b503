"""Multi-GPU layer: one process per GPU, one view (stereo pair / frame) per rank, ONE exchange step.

The reference is single-process / single-GPU (every device is the literal "CUDA:0"); the only parallel axis this
build adds is data-parallel sharding of independent views.  SGM, reprojection, voxel
down-sampling and normals need no communication.  Registration-to-a-common-view needs every rank to see view 0's
cloud, and the mesher (mesh_reconstruction.py, rank 0) needs the fused cloud: both are served by a single
all-gather-v of the per-view clouds over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).
On the fully connected xGMI mesh an all-gather moves each shard once per peer link; no all-reduce or ring is needed.
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


def all_gather_v(rows, device=None):
    """All-gather of row blocks with different row counts.  rows: float64 [n_r, c] numpy array of THIS rank.
    Returns the list of all ranks' blocks (rank order).  Two collectives: counts (world int64), then payload padded
    to the largest block.  Without an initialised process group it returns [rows] (world size 1)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2:
        raise ValueError("all_gather_v expects a 2-D array")
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return [rows]
    import torch
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    cnt = torch.tensor([rows.shape[0], rows.shape[1]], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [c.cpu().numpy() for c in cnts]
    cols = {int(c[1]) for c in cnts if c[0] > 0} or {rows.shape[1]}
    if len(cols) != 1:
        raise ValueError(f"ranks disagree on the column count: {sorted(cols)}")
    ncol = cols.pop()
    nmax = max(int(c[0]) for c in cnts)
    pad = torch.zeros((max(nmax, 1), ncol), dtype=torch.float64, device=device)
    if rows.shape[0]:
        pad[:rows.shape[0]] = torch.from_numpy(rows).to(device)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[:int(c[0])].cpu().numpy() for o, c in zip(out, cnts)]


def gather_rows_by_view(local, n_views, device=None):
    """local: {view_id: float64 [n, c]} owned by this rank.  Returns {view_id: array} for ALL views on every rank:
    one all_gather_v of the concatenated local blocks plus a tiny header (view id, row count) per block."""
    ids = sorted(local)
    ncol = next((local[v].shape[1] for v in ids), 0)
    hdr = np.array([[v, local[v].shape[0]] for v in ids], dtype=np.float64).reshape(-1, 2)
    hdrs = all_gather_v(hdr, device)
    body = np.concatenate([local[v] for v in ids], 0) if ids else np.zeros((0, ncol))
    widths = all_gather_v(np.array([[body.shape[1]]], dtype=np.float64), device)
    ncol = int(max(w[0, 0] for w in widths))
    if body.shape[1] != ncol:
        body = np.zeros((0, ncol))
    bodies = all_gather_v(body, device)
    out = {}
    for h, b in zip(hdrs, bodies):
        o = 0
        for v, n in h.astype(np.int64):
            out[int(v)] = b[o:o + n]
            o += n
    if sorted(out) != list(range(n_views)):
        raise RuntimeError(f"views missing after the exchange: have {sorted(out)}, want 0..{n_views - 1}")
    return out
