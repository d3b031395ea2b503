import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_project_amd")
ctx = r3d.default_context(0)
rows, row_bytes = 2448, 3136 * 256
for mode in (0, 1):
    for write in (0, 1):
        for delay in (0, 8, 24):
            ms = ctypes.c_float()
            ctx.call("r3d_debug_streambench", mode, rows, ctypes.c_uint64(row_bytes), write, delay, 5, ctypes.byref(ms))
            gb = rows * row_bytes * (2 if write else 1) / 1e9
            print(f"mode={'256B' if mode == 0 else '1KB '} write={write} delay={delay:2d}: {ms.value:7.3f} ms  {gb / ms.value:6.2f} TB/s", flush=True)
for rows2 in (612, 1224, 4896):
    ms = ctypes.c_float()
    ctx.call("r3d_debug_streambench", 0, rows2, ctypes.c_uint64(2448 * row_bytes // rows2), 1, 8, 5, ctypes.byref(ms))
    print(f"256B rw rows={rows2}: {ms.value:.3f} ms {2448 * row_bytes * 2 / 1e9 / ms.value:.2f} TB/s")

print("non-temporal variants (mode bit 1: nt loads, bit 2: nt stores), delay 8:")
for shape in (0, 1):
    for nt in (0, 2, 4, 6):
        for write in (0, 1):
            if not write and nt & 4:
                continue
            ms = ctypes.c_float()
            ctx.call("r3d_debug_streambench", shape | nt, rows, ctypes.c_uint64(row_bytes), write, 8, 5, ctypes.byref(ms))
            gb = rows * row_bytes * (2 if write else 1) / 1e9
            print(f"shape={'256B' if shape == 0 else '1KB '} ntload={int(bool(nt & 2))} ntstore={int(bool(nt & 4))} write={write}: {ms.value:7.3f} ms  {gb / ms.value:6.2f} TB/s", flush=True)
