"""On-disk formats the reference's scanner writes and re-reads (main.py:72,79; test/check84.py:161-186):
binary little-endian PLY with `double x,y,z[,nx,ny,nz]` + `uchar red,green,blue` ("Created by Open3D"),
16-bit PNG depth (millimetres) and the .npz calibration (Q matrix)."""
import numpy as np


def read_ply(path):
    """-> dict(points[, normals][, colors, colors_f][, faces]).  faces: int32 [M,3] when the file has a triangle list
    (`element face M` / `property list uchar uint vertex_indices`, what write_triangle_mesh produces, mesh_saving.py:15)."""
    types = {b"double": "<f8", b"float": "<f4", b"uchar": "u1", b"int": "<i4", b"uint": "<u4", b"short": "<i2", b"ushort": "<u2"}
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        props, n, nf, cur, face_list = [], 0, 0, None, None
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
                cur = tok[1]
                if cur == b"vertex":
                    n = int(tok[2])
                elif cur == b"face":
                    nf = int(tok[2])
            elif tok[0] == b"property" and cur == b"vertex":
                props.append((tok[2].decode(), types[tok[1]]))
            elif tok[0] == b"property" and cur == b"face" and tok[1] == b"list":
                face_list = (types[tok[2]], types[tok[3]])
        dt = np.dtype(props)
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        faces = None
        if nf and face_list:
            fdt = np.dtype([("n", face_list[0]), ("v", face_list[1], (3,))])
            raw = np.frombuffer(f.read(nf * fdt.itemsize), dtype=fdt, count=nf)
            if (raw["n"] != 3).any():
                raise ValueError("only triangle faces are supported")
            faces = raw["v"].astype(np.int32)
    out = {"points": np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float64)}
    if "nx" in dt.names:
        out["normals"] = np.stack([data["nx"], data["ny"], data["nz"]], 1).astype(np.float64)
    if "red" in dt.names:
        out["colors"] = np.stack([data["red"], data["green"], data["blue"]], 1)
        out["colors_f"] = out["colors"] / 255.0
    if faces is not None:
        out["faces"] = faces
    return out


def write_ply(path, points, normals=None, colors=None, faces=None):
    """colors: float in [0,1] (written as floor(c*255 + 0.5), the rounding verified on the recorded PLY files).
    faces: optional int [M,3] triangle list, written as `property list uchar uint vertex_indices`."""
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
    hdr += [f"property {names[t]} {nm}" for nm, t in fields]
    frec = None
    if faces is not None:
        fa = np.asarray(faces).reshape(-1, 3)
        if len(fa) and (fa.min() < 0 or fa.max() >= len(p)):
            raise ValueError("face index out of range")
        frec = np.zeros(len(fa), dtype=np.dtype([("n", "u1"), ("v", "<u4", (3,))]))
        frec["n"] = 3
        frec["v"] = fa
        hdr += [f"element face {len(fa)}", "property list uchar uint vertex_indices"]
    hdr += ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        f.write(rec.tobytes())
        if frec is not None:
            f.write(frec.tobytes())


def read_depth_png(path):
    from PIL import Image
    return np.asarray(Image.open(path)).astype(np.uint16)


def load_calibration(path):
    """Calib_depth/depth2.py:35-67 getStereoCameraParameters: the .npz keys, Q included."""
    d = np.load(path)
    return {k: d[k] for k in d.files}
