import ctypes, importlib, os, sys
sys.path.insert(0, os.getcwd())
r3d = importlib.import_module("3d_reconstruction_project_amd")
ctx = r3d.default_context(0)
total = 2448 * 3136 * 256
for mode in (0, 1):
    for write in (0, 1):
        for rows in (612, 1224, 2448, 4896, 9792, 19584, 39168):
            ms = ctypes.c_float()
            ctx.call("r3d_debug_streambench", mode, rows, ctypes.c_uint64(total // rows), write, 0, 5, ctypes.byref(ms))
            print(f"mode={'256B' if mode == 0 else '1KB '} write={write} waves={rows:6d}: {ms.value:7.3f} ms {total * (2 if write else 1) / 1e9 / ms.value:6.2f} TB/s", flush=True)
