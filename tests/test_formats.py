"""On-disk formats either side of the path: host C++ behind the C ABI, checked
against hand-made fixtures in the reference's formats (tests/golden/formats) and a pure-Python restatement of
the stream formatting rules (calibration.cpp:22-79, camera.hpp:19-72,138-180, configuration.hpp:31-91).
yaml-cpp is not in the image, so the YAML emitter's exact text is "parity unpinned"; the reader is checked on
yaml-cpp-style block output, on flow sequences and against PyYAML's reading of the same files."""
import os

import numpy as np
import pytest
import yaml

from cvx_amd import formats
from cvx_amd.capi import CvxError

G = os.path.join(os.path.dirname(__file__), "golden", "formats")


def g6(v):
    """operator<< of a float at the stream's default precision (6 significant digits, %g)"""
    return "%g" % float(np.float32(v))


def test_calib_data_load_fixture():
    d = formats.load_calib_data(os.path.join(G, "calib_small.data"))
    assert (d["width"], d["height"]) == (640, 480)
    np.testing.assert_array_equal(d["coords"], [[0, 0, 0], [25, 0, 0], [0, 25, 0], [25, 25, 0]])
    np.testing.assert_array_equal(d["view_off"], [0, 4, 4, 6])        # the middle view has no markers
    np.testing.assert_array_equal(d["point_id"], [0, 1, 2, 3, 3, 1])
    assert d["image_paths"] == ["img/view_000.png", "img/view_001.png", "img/view_002.png"]
    # values pass through float (cv::Point2f)
    np.testing.assert_array_equal(d["obs_x"], np.float32([100.5, 150.125, 99.75, 151, 320.333, 1e-05]).astype(np.float64))
    np.testing.assert_array_equal(d["obs_y"], np.float32([200.25, 201, 251.5, 250, 240.667, 123457]).astype(np.float64))


def test_calib_data_save_is_byte_identical_to_fixture(tmp_path):
    src = os.path.join(G, "calib_small.data")
    d = formats.load_calib_data(src)
    out = tmp_path / "out.data"
    formats.save_calib_data(out, d["width"], d["height"], d["coords"], d["image_paths"], d["view_off"],
                            d["point_id"], d["obs_x"], d["obs_y"])
    assert out.read_bytes() == open(src, "rb").read()


def test_calib_data_save_matches_stream_formatting(tmp_path):
    rng = np.random.default_rng(5)
    P, V = 12, 7
    coords = rng.uniform(-100, 100, (P, 3))
    counts = rng.integers(0, P + 1, V); counts[2] = 0
    view_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    pid = np.concatenate([rng.permutation(P)[:c] for c in counts]).astype(np.int32)
    ox = rng.uniform(0, 2000, pid.size) * 10.0 ** rng.integers(-6, 3, pid.size)
    oy = rng.uniform(0, 2000, pid.size)
    paths = ["/data/run/%04d.png" % i for i in range(V)]
    out = tmp_path / "rand.data"
    formats.save_calib_data(out, 1920, 1080, coords, paths, view_off, pid, ox, oy)
    lines = ["1920 1080", str(P)] + ["%s %s %s" % tuple(g6(v) for v in c) for c in coords] + [str(V)]
    for v in range(V):
        lines += [paths[v], str(counts[v])]
        lines += ["%s %s %d" % (g6(ox[o]), g6(oy[o]), pid[o]) for o in range(view_off[v], view_off[v + 1])]
    assert out.read_text() == "\n".join(lines) + "\n"
    # reload: what survives is exactly what a 6-significant-digit text holds
    d = formats.load_calib_data(out)
    np.testing.assert_array_equal(d["view_off"], view_off)
    np.testing.assert_array_equal(d["point_id"], pid)
    np.testing.assert_allclose(d["obs_x"], ox, rtol=1e-5)
    np.testing.assert_allclose(d["coords"], coords, rtol=1e-5, atol=1e-4)
    # and the reloaded file is a fixed point of save
    out2 = tmp_path / "rand2.data"
    formats.save_calib_data(out2, d["width"], d["height"], d["coords"], d["image_paths"], d["view_off"],
                            d["point_id"], d["obs_x"], d["obs_y"])
    assert out2.read_bytes() == out.read_bytes()


def test_calib_data_empty_and_errors(tmp_path):
    out = tmp_path / "empty.data"
    formats.save_calib_data(out, 10, 20, np.zeros((0, 3)), [], [0], [], [], [])
    assert out.read_text() == "10 20\n0\n0\n"
    d = formats.load_calib_data(out)
    assert d["coords"].shape == (0, 3) and list(d["view_off"]) == [0] and d["image_paths"] == []
    with pytest.raises(CvxError):
        formats.load_calib_data(tmp_path / "nope.data")
    bad = tmp_path / "trunc.data"
    bad.write_text("640 480\n1\n0 0 0\n1\nimg.png\n3\n1 2 0\n")       # promises 3 markers, holds 1
    with pytest.raises(CvxError):
        formats.load_calib_data(bad)


def test_loaded_data_feeds_the_problem_description():
    """the loader hands out exactly the flat arrays cvx_calib_desc takes (calibration.cpp:385-402)"""
    d = formats.load_calib_data(os.path.join(G, "calib_small.data"))
    assert d["view_off"].dtype == np.int32 and d["point_id"].dtype == np.int32
    assert d["obs_x"].dtype == np.float64 and d["coords"].dtype == np.float64
    assert d["view_off"][-1] == d["point_id"].size == d["obs_x"].size == d["obs_y"].size
    assert d["point_id"].max() < d["coords"].shape[0]


@pytest.mark.parametrize("name", ["camera.yaml", "camera_flow.yaml"])
def test_camera_yaml_read_agrees_with_pyyaml(name):
    p = os.path.join(G, name)
    w, h, K, dist = formats.read_camera_yaml(p)
    y = yaml.safe_load(open(p))
    M = np.array(y["camera_matrix"]["data"], float).reshape(3, 3)       # row-major, camera.hpp:45-66
    assert (w, h) == (y["image_width"], y["image_height"])
    np.testing.assert_array_equal(K, [M[0, 0], M[1, 1], M[0, 2], M[1, 2]])
    np.testing.assert_array_equal(dist, np.array(y["distortion_coefficients"]["data"], float))


def test_camera_yaml_write_roundtrip_and_layout(tmp_path):
    K = np.array([525.5, 524.25, 319.5, 239.5]); dist = np.array([0.1, -0.25, 0.001, -0.002, 0.05])
    out = tmp_path / "cam.yaml"
    formats.write_camera_yaml(out, 640, 480, K, dist)
    # doubles carry 17 significant digits, so the round trip is exact
    w, h, K2, d2 = formats.read_camera_yaml(out)
    assert (w, h) == (640, 480)
    np.testing.assert_array_equal(K2, K); np.testing.assert_array_equal(d2, dist)
    y = yaml.safe_load(open(out))
    assert list(y) == ["image_width", "image_height", "camera_matrix", "distortion_coefficients"]
    assert (y["camera_matrix"]["rows"], y["camera_matrix"]["cols"]) == (3, 3)
    assert (y["distortion_coefficients"]["rows"], y["distortion_coefficients"]["cols"]) == (5, 1)
    np.testing.assert_array_equal(np.array(y["camera_matrix"]["data"], float),
                                  [525.5, 0, 319.5, 0, 524.25, 239.5, 0, 0, 1])
    # same text as the block-style fixture
    assert out.read_text() == open(os.path.join(G, "camera.yaml")).read()


def test_camera_yaml_errors(tmp_path):
    p = tmp_path / "bad.yaml"
    p.write_text("image_width: 640\nimage_height: 480\ncamera_matrix:\n  rows: 3\n  cols: 3\n  data: [1, 2]\n"
                 "distortion_coefficients:\n  rows: 5\n  cols: 1\n  data: [0,0,0,0,0]\n")
    with pytest.raises(CvxError):
        formats.read_camera_yaml(p)
    p.write_text("image_width: 640\n")
    with pytest.raises(CvxError):
        formats.read_camera_yaml(p)


def quat_R(qw, qx, qy, qz):
    n = np.sqrt(qw * qw + qx * qx + qy * qy + qz * qz)
    w, x, y, z = qw / n, qx / n, qy / n, qz / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_robot_config_quaternion():
    c = formats.read_robot_config(os.path.join(G, "robot_quat.yaml"))
    assert (c["base_frame"], c["tool_frame"], c["camera_frame"]) == ("base_link", "right_tool0", "xtion2/rgb_optical_frame")
    np.testing.assert_array_equal(c["base_to_tag"], [0.5, -0.25, 0.125])
    X = c["initial_cam"].reshape(4, 4).T
    np.testing.assert_allclose(X[:3, :3], quat_R(1.6, 0.2, -0.4, 0.4), atol=1e-15)      # q.normalized()
    np.testing.assert_array_equal(X[:3, 3], [0.1, -0.2, 0.3])
    np.testing.assert_array_equal(X[3], [0, 0, 0, 1])


def test_robot_config_roll_pitch_yaw():
    c = formats.read_robot_config(os.path.join(G, "robot_rpy.yaml"))
    r, p, y = 0.02, 0.48, 1.62
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    X = c["initial_cam"].reshape(4, 4).T
    np.testing.assert_allclose(X[:3, :3], Rx @ Ry @ Rz, atol=1e-15)
    np.testing.assert_array_equal(X[:3, 3], [-0.04, -0.01, 0.53])


def test_robot_config_missing_keys(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text('base_frame: "b"\ntool_frame: "t"\ncamera_frame: "c"\nbase_to_tag_x: 0\nbase_to_tag_y: 0\n'
                 "base_to_tag_z: 0\ninitial_cam_estimation:\n  x: 0\n  y: 0\n  z: 0\n")
    with pytest.raises(CvxError):
        formats.read_robot_config(p)
    p.write_text('base_frame: "b"\n')
    with pytest.raises(CvxError):
        formats.read_robot_config(p)


def test_launch_file(tmp_path):
    X = np.eye(4); X[:3, :3] = quat_R(0.5, 0.5, -0.5, 0.5); X[:3, 3] = [0.123456789, -2.5, 1e-7]
    out = tmp_path / "cam.launch"
    formats.write_launch_file(out, "right_tool0", "xtion2/rgb_optical_frame", X.T.reshape(-1))
    txt = out.read_text().splitlines()
    assert txt[0] == '<?xml version="1.0"?>' and txt[1] == "<launch>" and txt[3] == "</launch>"
    # node name = last four characters of the tool frame; numbers at the stream's default precision
    assert txt[2] == ('<node name="ool0_to_camera" pkg="tf" type="static_transform_publisher" args="'
                      '0.123457 -2.5 1e-07 0.5 -0.5 0.5 0.5 right_tool0 xtion2/rgb_optical_frame 100"/>')
