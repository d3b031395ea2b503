"""Infinity Cache probe: the stream kernel of tools/gpu_streambench.py re-run over the SAME buffers with footprints from 32 MB to
2 GB (2448 waves, 1 KB requests).  A footprint that stays resident in the 256 MiB Infinity Cache is re-read on-die by the next
launch; larger ones come from HBM every time.  Tells whether column-slab pipelining of the SGM kernels (cost -> hscan -> vscan on
slabs small enough to stay resident) can take volume re-reads off HBM.  Run on the GPU box."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
ctx = r3d.default_context(0)
rows = 2448
for write in (0, 1):
    for mb in (32, 64, 96, 128, 192, 256, 384, 512, 1024, 1966):
        row_bytes = (mb * (1 << 20) // rows) // 4096 * 4096
        ms = ctypes.c_float()
        ctx.call("r3d_debug_streambench", 1, rows, ctypes.c_uint64(row_bytes), write, 0, 20, ctypes.byref(ms))
        moved = rows * row_bytes * (2 if write else 1)
        print(f"write={write} buffer={rows * row_bytes / 2**20:7.1f} MiB (footprint {moved / 2**20:7.1f} MiB): {ms.value * 1e3:8.1f} us  {moved / 1e9 / ms.value:6.2f} TB/s", flush=True)
