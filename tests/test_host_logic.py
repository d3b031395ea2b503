"""CPU tests of the host-side mirror of the reference interfaces (no GPU, no compute calls)."""
import os

import numpy as np
import pytest

from oracle import cloud_oracle as co
from tests.conftest import GOLDEN


def test_ply_reader_writer_roundtrip_matches_recorded_file(r3d, tmp_path):
    src = os.path.join(GOLDEN, "output84/pcd_00008.ply")
    d = r3d.io_formats.read_ply(src)
    ref = co.read_ply(src)
    np.testing.assert_array_equal(d["points"], ref["points"])
    np.testing.assert_array_equal(d["normals"], ref["normals"])
    out = tmp_path / "rt.ply"
    r3d.io_formats.write_ply(str(out), d["points"], d["normals"], d["colors_f"])
    assert open(out, "rb").read() == open(src, "rb").read()            # byte-identical to what Open3D wrote


def test_depth_png_and_calibration_loaders(r3d):
    d = r3d.io_formats.read_depth_png(os.path.join(GOLDEN, "output84/depth_00008.png"))
    assert d.dtype == np.uint16 and d.shape == (480, 640)
    cal = r3d.io_formats.load_calibration(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))
    assert cal["Q"].shape == (4, 4) and abs(1.0 / cal["Q"][3, 2] - 31.5) < 0.1      # baseline in mm
    Q = r3d.pipeline.scaled_Q(cal["Q"], 3.4, unit=1e-3)
    assert np.isclose(Q[2, 3], cal["Q"][2, 3] * 3.4e-3) and Q[3, 2] == cal["Q"][3, 2]


def test_stereo_object_protocol(r3d):
    """cv2-style accessors used by the key handlers of depth1.py:240-265 and createRightMatcher's parameter mapping."""
    m = r3d.reference_matcher(numDisparities=128, blockSize=5, family="depth2")
    assert (m.getNumDisparities(), m.getBlockSize(), m.getP1(), m.getP2(), m.getUniquenessRatio()) == (128, 5, 600, 2400, 15)
    m.setBlockSize(7)
    m.setNumDisparities(64)
    m.setSpeckleWindowSize(50)
    assert (m.getBlockSize(), m.getNumDisparities(), m.getSpeckleWindowSize(), m.getMode()) == (7, 64, 50, 2)
    r = r3d.createRightMatcher(m)
    assert (r.getMinDisparity(), r.getNumDisparities(), r.getUniquenessRatio(), r.getDisp12MaxDiff(), r.getSpeckleWindowSize()) == \
        (-63, 64, 0, 1000000, 0)
    m4 = r3d.reference_matcher(family="depth4")
    assert (m4.getUniquenessRatio(), m4.getSpeckleWindowSize(), m4.getSpeckleRange()) == (10, 50, 32)
    p = m.params_struct()
    assert (p.numDisparities, p.blockSize, p.mode) == (64, 7, 2)
    with pytest.raises(AttributeError):
        m.getNoSuchThing()
    with pytest.raises(ValueError):
        r3d.reference_matcher(family="depth9")


def test_device_string_parsing(r3d):
    assert r3d._lib.parse_device("CUDA:0") == 0 and r3d._lib.parse_device("HIP:3") == 3 and r3d._lib.parse_device(2) == 2
    with pytest.raises(r3d.R3DError):
        r3d._lib.parse_device("CPU:0")                                   # no CPU backend exists


def test_cloud_container_interop(r3d):
    class Vec(list):                                                     # stand-in for o3d.utility.Vector3dVector
        pass

    class Foreign:                                                       # stand-in for a legacy o3d PointCloud
        def __init__(self):
            self.points, self.colors, self.normals = Vec(), Vec(), Vec()
    f = Foreign()
    f.points = Vec([[0, 0, 0], [1, 2, 3]])
    p, c, n = r3d.pointcloud.as_arrays(f)
    assert p.shape == (2, 3) and c is None and n is None
    out = r3d.pointcloud.like(f, np.ones((3, 3)), normals=np.zeros((3, 3)))
    assert isinstance(out, Foreign) and isinstance(out.points, Vec) and len(out.points) == 3
    assert isinstance(r3d.pointcloud.like(np.zeros((1, 3)), np.ones((2, 3))), r3d.PointCloud)


def test_fuse_loop_skips_empty_frames_without_touching_the_gpu(r3d):
    """main.py:39,53-54: empty / failed captures are skipped; the first valid frame initialises the model."""
    first = r3d.PointCloud(np.random.default_rng(0).random((10, 3)), colors=np.ones((10, 3)))
    model = r3d.pipeline.fuse([None, r3d.PointCloud(), first, None])
    assert len(model) == 10 and model.has_colors() and not model.has_normals()


def test_ply_triangle_mesh_round_trip(r3d, tmp_path):
    """mesh_saving.py:15 writes the reconstructed mesh with write_triangle_mesh: vertices + `list uchar uint vertex_indices`."""
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(50, 3))
    nrm = rng.normal(size=(50, 3))
    col = rng.random((50, 3))
    faces = rng.integers(0, 50, (80, 3))
    f = tmp_path / "mesh.ply"
    r3d.io_formats.write_ply(str(f), pts, nrm, col, faces=faces)
    head = f.read_bytes().split(b"end_header")[0].decode()
    assert "element face 80" in head and "property list uchar uint vertex_indices" in head
    back = r3d.io_formats.read_ply(str(f))
    np.testing.assert_array_equal(back["points"], pts)
    np.testing.assert_array_equal(back["normals"], nrm)
    np.testing.assert_array_equal(back["faces"], faces)
    np.testing.assert_array_equal(back["colors"], np.floor(col * 255 + 0.5).astype(np.uint8))
    with pytest.raises(ValueError):
        r3d.io_formats.write_ply(str(f), pts, faces=[[0, 1, 50]])
