"""numpy-facing wrappers of the on-disk formats either side of the path:
calib.data (calibration.cpp:22-79), PinholeCamera YAML (camera.hpp:138-180), CameraRobotConfiguration
(configuration.hpp:31-91).  Every function is one C-ABI call into libcvx_amd.so (host code only)."""
import ctypes as C
import os

import numpy as np

from .capi import check, f64, i32, lib, pd, pi


def load_calib_data(path):
    """-> dict(width, height, coords[P,3], view_off[V+1], point_id[N], obs_x[N], obs_y[N], image_paths[V])"""
    h = C.c_void_p()
    check(lib().cvx_calib_data_load(os.fsencode(path), C.byref(h)), "cvx_calib_data_load")
    try:
        w, ht, nc, nv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        no = C.c_int64()
        lib().cvx_calib_data_info(h, C.byref(w), C.byref(ht), C.byref(nc), C.byref(nv), C.byref(no))
        coords = np.zeros((nc.value, 3)); view_off = np.zeros(nv.value + 1, np.int32)
        point_id = np.zeros(no.value, np.int32); ox = np.zeros(no.value); oy = np.zeros(no.value)
        lib().cvx_calib_data_copy(h, pd(coords), pi(view_off), pi(point_id), pd(ox), pd(oy))
        paths = [lib().cvx_calib_data_path(h, i).decode() for i in range(nv.value)]
    finally:
        lib().cvx_calib_data_free(h)
    return dict(width=w.value, height=ht.value, coords=coords, view_off=view_off, point_id=point_id,
                obs_x=ox, obs_y=oy, image_paths=paths)


def save_calib_data(path, width, height, coords, image_paths, view_off, point_id, obs_x, obs_y):
    coords = f64(np.asarray(coords).reshape(-1, 3)); view_off = i32(view_off); point_id = i32(point_id)
    obs_x, obs_y = f64(obs_x), f64(obs_y)
    arr = (C.c_char_p * max(len(image_paths), 1))(*[os.fsencode(p) for p in image_paths])
    check(lib().cvx_calib_data_save(os.fsencode(path), int(width), int(height), coords.shape[0], pd(coords),
                                    len(image_paths), arr, pi(view_off), pi(point_id), pd(obs_x), pd(obs_y)),
          "cvx_calib_data_save")


def read_camera_yaml(path):
    """-> (width, height, K = [fx, fy, cx, cy], dist = [k1, k2, p1, p2, k3])"""
    w, h = C.c_int(), C.c_int()
    K = np.zeros(4); d = np.zeros(5)
    check(lib().cvx_camera_yaml_read(os.fsencode(path), C.byref(w), C.byref(h), pd(K), pd(d)), "cvx_camera_yaml_read")
    return w.value, h.value, K, d


def write_camera_yaml(path, width, height, K, dist):
    K, dist = f64(K), f64(dist)
    check(lib().cvx_camera_yaml_write(os.fsencode(path), int(width), int(height), pd(K), pd(dist)),
          "cvx_camera_yaml_write")


def read_robot_config(path):
    """-> dict(base_frame, tool_frame, camera_frame, base_to_tag[3], initial_cam[16] column-major)"""
    cap = 256
    b, t, c = (C.create_string_buffer(cap) for _ in range(3))
    tag = np.zeros(3); X = np.zeros(16)
    check(lib().cvx_robot_config_read(os.fsencode(path), b, t, c, cap, pd(tag), pd(X)), "cvx_robot_config_read")
    return dict(base_frame=b.value.decode(), tool_frame=t.value.decode(), camera_frame=c.value.decode(),
                base_to_tag=tag, initial_cam=X)


def write_launch_file(path, tool_frame, camera_frame, X):
    X = f64(X)
    check(lib().cvx_robot_config_write_launch(os.fsencode(path), tool_frame.encode(), camera_frame.encode(), pd(X)),
          "cvx_robot_config_write_launch")
