"""On-disk formats the reference's scanner writes and re-reads (main.py:72,79; test/check84.py:161-186):
binary little-endian PLY with `double x,y,z[,nx,ny,nz]` + `uchar red,green,blue` ("Created by Open3D"),
16-bit PNG depth (millimetres) and the .npz calibration (Q matrix)."""
import numpy as np


def read_ply(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        props, n, in_vertex = [], 0, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("truncated PLY header")
            tok = line.strip().split()
            if not tok:
                continue
            if tok[0] == b"end_header":
                break
            if tok[0] == b"format" and tok[1] != b"binary_little_endian":
                raise ValueError("only binary_little_endian PLY is supported")
            if tok[0] == b"element":
                in_vertex = tok[1] == b"vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == b"property" and in_vertex:
                props.append((tok[2].decode(), {b"double": "<f8", b"float": "<f4", b"uchar": "u1", b"int": "<i4"}[tok[1]]))
        dt = np.dtype(props)
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    out = {"points": np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float64)}
    if "nx" in dt.names:
        out["normals"] = np.stack([data["nx"], data["ny"], data["nz"]], 1).astype(np.float64)
    if "red" in dt.names:
        out["colors"] = np.stack([data["red"], data["green"], data["blue"]], 1)
        out["colors_f"] = out["colors"] / 255.0
    return out


def write_ply(path, points, normals=None, colors=None):
    """colors: float in [0,1] (written as floor(c*255 + 0.5), the rounding verified on the recorded PLY files)."""
    p = np.asarray(points, np.float64).reshape(-1, 3)
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    if normals is not None and len(normals):
        fields += [("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")]
    if colors is not None and len(colors):
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.zeros(len(p), dtype=np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = p[:, 0], p[:, 1], p[:, 2]
    if "nx" in rec.dtype.names:
        q = np.asarray(normals, np.float64).reshape(-1, 3)
        rec["nx"], rec["ny"], rec["nz"] = q[:, 0], q[:, 1], q[:, 2]
    if "red" in rec.dtype.names:
        c = np.clip(np.floor(np.asarray(colors, np.float64).reshape(-1, 3) * 255.0 + 0.5), 0, 255).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    names = {"<f8": "double", "u1": "uchar"}
    hdr = ["ply", "format binary_little_endian 1.0", "comment Created by Open3D", f"element vertex {len(p)}"]
    hdr += [f"property {names[t]} {nm}" for nm, t in fields] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        f.write(rec.tobytes())


def read_depth_png(path):
    from PIL import Image
    return np.asarray(Image.open(path)).astype(np.uint16)


def load_calibration(path):
    """Calib_depth/depth2.py:35-67 getStereoCameraParameters: the .npz keys, Q included."""
    d = np.load(path)
    return {k: d[k] for k in d.files}
