"""Multi-GPU layer: one process per GPU, one view (stereo pair / frame) per rank, ONE exchange step.

The reference is single-process / single-GPU (every device is the literal "CUDA:0"); the only parallel axis this
build adds is data-parallel sharding of independent views.  SGM, reprojection, voxel
down-sampling and normals need no communication.  Registration-to-a-common-view needs every rank to see view 0's
cloud, and the mesher (mesh_reconstruction.py:22-37, rank 0) needs the fused cloud: both are served by a single
all-gather-v of the per-view clouds over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).
On the fully connected xGMI mesh an all-gather moves each shard once per peer link; no all-reduce or ring is needed.

The payload never leaves HBM under "nccl": clouds are torch tensors that wrap device memory the HIP library wrote
(r3d_disparity_to_cloud_resident), the collective runs on torch's current stream, which init() / shared_stream() also make the
library's stream, and the gathered blocks are handed back to the library as device pointers (r3d_icp_dev,
r3d_transform_points_dev).  Only the row counts (a few int64 per rank) and the 4x4 transforms are read on the host.

Stream contract.  A library Context enqueues on ITS stream (own, non-blocking: not ordered against torch's null stream).  Code
that mixes torch ops / collectives with library calls on the same buffers must make the two ONE stream: share_stream(ctx) does
it for the rest of the process (init() calls it), `with shared_stream(ctx):` for one block.  Neither ever hands the library
torch's default-stream handle 0 -- r3d_set_stream reads NULL as "back to the own stream" -- when torch is on its null stream a
side stream is made current instead and the null stream is joined to it with events on both ends.
"""
import contextlib
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


def share_stream(ctx):
    """Process-wide: torch's current stream on ctx.device and the library context's stream become ONE stream, so library
    kernels, RCCL collectives and torch ops are ordered by enqueue order alone.  If torch is on its default (null) stream a new
    torch stream is created, ordered after whatever the null stream holds, and made torch's current stream (handle 0 cannot be
    handed to r3d_set_stream: NULL means "own stream" there).  Returns the torch stream; the context keeps it alive."""
    import torch
    cur = torch.cuda.current_stream(ctx.device)
    if cur.cuda_stream == 0:
        s = torch.cuda.Stream(device=ctx.device)
        s.wait_stream(cur)
        torch.cuda.set_stream(s)
        cur = s
    prev = ctx.get_stream()
    if prev != cur.cuda_stream:
        cur.wait_stream(torch.cuda.ExternalStream(prev, device=ctx.device))     # library work already queued stays ahead
        ctx.set_stream(cur.cuda_stream)
    ctx._torch_stream = cur
    return cur


@contextlib.contextmanager
def shared_stream(ctx):
    """`with shared_stream(ctx) as s:` -- inside the block torch's current stream and the context's stream are the same stream
    `s`; on exit both are what they were, and each side is ordered after the block (events, no host synchronisation).  Tensors
    allocated inside and used after the block on another stream should be given record_stream(torch.cuda.current_stream())."""
    import torch
    outer = torch.cuda.current_stream(ctx.device)
    prev = ctx.get_stream()
    if prev == outer.cuda_stream and prev != 0:
        yield outer
        return
    lib_prev = torch.cuda.ExternalStream(prev, device=ctx.device)
    inner = outer if outer.cuda_stream != 0 else torch.cuda.Stream(device=ctx.device)
    if inner is not outer:
        inner.wait_stream(outer)
    inner.wait_stream(lib_prev)
    ctx.set_stream(inner.cuda_stream)
    try:
        with torch.cuda.stream(inner):
            yield inner
    finally:
        ctx.set_stream(prev)
        lib_prev.wait_stream(inner)
        if inner is not outer:
            outer.wait_stream(inner)


def init(backend=None, ctx=None, timeout_s=None):
    """Binds this rank to its GPU BEFORE any other GPU call and returns (rank, world, ctx): torch.cuda.set_device(LOCAL_RANK),
    a library Context on the same device sharing ONE stream with torch (share_stream: library kernels, RCCL collectives and
    torch ops are ordered without events), and the process group ("nccl" when a GPU is visible, else "gloo") if
    WORLD_SIZE > 1 (or R3D_FORCE_DIST is set, to rehearse the collective path with one rank).  Idempotent.
    timeout_s: collective timeout of the group (a rank that dies must not leave its peers waiting for the default 10 minutes)."""
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
        share_stream(ctx)
    import torch.distributed as dist
    if (world > 1 or os.environ.get("R3D_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        if timeout_s:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
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


class RemoteStageError(RuntimeError):
    """A rank-local stage failed on SOME rank; raised on EVERY rank at the same point (after the collective that carried the
    failure flag), so no rank is left waiting inside a later collective."""


def gather_transforms(mine, n_views):
    """mine: {view_id: 4x4 numpy, or an Exception if that view's registration failed on this rank} of the owned views ->
    {view_id: 4x4 numpy} of all views on every rank (one small collective).  A failed view travels as a flag in its slot: every
    rank then raises RemoteStageError together instead of the healthy ranks hanging in the next collective."""
    import torch
    d = _dist()
    world = d.get_world_size() if d is not None else 1
    slots = (n_views + world - 1) // world
    buf = torch.zeros((slots, 18), dtype=torch.float64)
    buf[:, 0] = -1
    for i, v in enumerate(sorted(mine)):
        buf[i, 0] = v
        if isinstance(mine[v], BaseException):
            buf[i, 17] = 1.0
        else:
            buf[i, 1:17] = torch.from_numpy(np.asarray(mine[v], dtype=np.float64).reshape(16))
    allb = _all_gather_fixed(buf.to(exchange_device())).cpu().numpy().reshape(-1, 18)
    failed = sorted(int(r[0]) for r in allb if r[0] >= 0 and r[17] != 0)
    if failed:
        own = "; ".join(f"view {v}: {type(e).__name__}: {e}" for v, e in sorted(mine.items()) if isinstance(e, BaseException))
        raise RemoteStageError(f"registration failed for view(s) {failed}" + (f" [this rank: {own}]" if own else ""))
    out = {int(r[0]): r[1:17].reshape(4, 4).copy() for r in allb if r[0] >= 0}
    if sorted(out) != list(range(n_views)):
        raise RuntimeError(f"transforms missing after the exchange: have {sorted(out)}")
    return out


def agree(ok, what="stage"):
    """Every rank passes whether its rank-local stage succeeded (`ok`: True or the exception it caught); returns normally on all
    ranks if all succeeded, raises RemoteStageError on ALL ranks otherwise (one tiny collective).  Call it between a stage that
    can fail locally and the first collective that follows."""
    import torch
    good = ok is True
    d = _dist()
    if d is not None and (d.get_world_size() > 1 or os.environ.get("R3D_FORCE_DIST")):
        flags = _all_gather_fixed(torch.tensor([0.0 if good else 1.0], dtype=torch.float64, device=exchange_device())).cpu().numpy().reshape(-1)
    else:
        flags = np.array([0.0 if good else 1.0])
    if flags.any():
        bad = [int(r) for r in np.nonzero(flags)[0]]
        own = "" if good else f" [this rank: {type(ok).__name__}: {ok}]"
        raise RemoteStageError(f"{what} failed on rank(s) {bad}{own}")


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
