"""ctypes binding of the C ABI in include/rna.h (librna.so, built in-tree by csrc/Makefile).

This is plumbing for tests and bench.py: numpy structured arrays mirror the C structs one to one,
host-pointer calls take numpy arrays, `_device` calls take raw device pointers (e.g.
``torch.Tensor.data_ptr()``).  There is NO CPU fallback: if librna.so is missing or no gfx950
device is visible the import / engine creation fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("RNA_LIB") or "librna.so")   # RNA_LIB: developer switch to an alternative build

RNA_OK = 0
ABI_VERSION = 5   # include/rna.h: RNA_ABI_VERSION
STATUS = {0: "RNA_OK", -1: "RNA_EINVAL", -2: "RNA_ENOMEM", -3: "RNA_EHIP", -4: "RNA_ECAPACITY",
          -5: "RNA_ESTATE", -6: "RNA_ENODEVICE"}
LAYER_MASTER, LAYER_LASER, LAYER_RANGE = 0, 1, 2
KERNELS = ["himm_prep", "himm_raster", "himm_apply", "compose_master", "nbr_mask", "vfh_step",
           "astar_search", "astar_init", "rrt", "to_occupancy_grid", "astar_reset"]

# every symbol include/rna.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "rna_create", "rna_destroy", "rna_last_error", "rna_abi_version", "rna_get_geometry",
    "rna_layer_upload", "rna_layer_download", "rna_layer_fill", "rna_layer_device_ptr", "rna_stream",
    "rna_synchronize", "rna_synchronize_map", "rna_hw_queue_advice", "rna_get_index", "rna_get_position", "rna_geometry_index", "rna_geometry_position", "rna_line_cells",
    "rna_circle_cells", "rna_submap_cells", "rna_clone",
    "rna_himm_update", "rna_himm_update_device", "rna_compose_master", "rna_update_map",
    "rna_update_map_device", "rna_move", "rna_himm_set_window", "rna_layer_pack_region", "rna_layer_unpack_region", "rna_last_dirty_tiles", "rna_layer_pack_tiles",
    "rna_layers_unpack_tiles", "rna_layer_unpack_region_tracked", "rna_last_dirty_tiles_device", "rna_layer_pack_tiles_device",
    "rna_layers_unpack_tiles_device",
    "rna_vfh_default_params", "rna_vfh_init", "rna_vfh_reset", "rna_vfh_hist_size", "rna_vfh_step_batch",
    "rna_vfh_step_batch_device", "rna_vfh_update_batch",
    "rna_astar_configure", "rna_astar_set_pipeline_depth", "rna_astar_set_page_cap", "rna_astar_effective_config", "rna_astar_batch", "rna_astar_batch_device", "rna_astar_settled_counts", "rna_astar_job_counters",
    "rna_astar_download_nbr_mask",
    "rna_graph_astar_batch", "rna_rrt_batch", "rna_rrt_batch_device",
    "rna_to_occupancy_grid", "rna_to_occupancy_grid_device", "rna_from_occupancy_grid", "rna_vfh_hist_msg_batch",
    "rna_tailor_plan", "rna_follow_plan", "rna_get_submap", "rna_get_submap_device", "rna_create_submap", "rna_scan_to_rays", "rna_scan_to_rays_device", "rna_scan_projected_beams", "rna_range_to_rays",
    "rna_scan_to_rays_tf", "rna_scan_to_rays_tf_device", "rna_range_to_rays_tf",
    "rna_profile_enable", "rna_profile_reset", "rna_profile_get", "rna_kernel_name",
]


class Geometry(C.Structure):
    _fields_ = [("length", C.c_double * 2), ("position", C.c_double * 2), ("resolution", C.c_double),
                ("size", C.c_int32 * 2), ("start_index", C.c_int32 * 2)]


class SubmapInfo(C.Structure):
    _fields_ = [("length", C.c_double * 2), ("position", C.c_double * 2), ("size", C.c_int32 * 2),
                ("top_left", C.c_int32 * 2)]


class VfhParams(C.Structure):
    _fields_ = [("cell_size", C.c_double), ("window_diameter", C.c_int32), ("sector_angle", C.c_int32),
                ("safety_dist_0ms", C.c_double), ("safety_dist_1ms", C.c_double),
                ("max_speed", C.c_int32), ("max_speed_narrow_opening", C.c_int32),
                ("max_speed_wide_opening", C.c_int32), ("max_acceleration", C.c_int32),
                ("min_turnrate", C.c_int32), ("max_turnrate_0ms", C.c_int32), ("max_turnrate_1ms", C.c_int32),
                ("min_turn_radius_safety_factor", C.c_double),
                ("free_space_cutoff_0ms", C.c_double), ("obs_cutoff_0ms", C.c_double),
                ("free_space_cutoff_1ms", C.c_double), ("obs_cutoff_1ms", C.c_double),
                ("weight_desired_dir", C.c_double), ("weight_current_dir", C.c_double),
                ("robot_radius", C.c_double)]


RAY_DTYPE = np.dtype([("sx", "<f8"), ("sy", "<f8"), ("ex", "<f8"), ("ey", "<f8"),
                      ("clear_end", "<i4"), ("_pad", "<i4")])
POSE_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("yaw", "<f8"), ("dt", "<f8"),
                       ("current_speed", "<i4"), ("goal_direction", "<f4"), ("goal_distance", "<f4"),
                       ("goal_tolerance", "<f4")])
VFH_OUT_DTYPE = np.dtype([("chosen_speed", "<i4"), ("chosen_turnrate", "<i4"), ("picked_angle", "<f4"),
                          ("emergency", "<i4")])
SCAN_DTYPE = np.dtype([("angle_min", "<f4"), ("angle_max", "<f4"), ("angle_increment", "<f4"), ("range_min", "<f4"),
                       ("range_max", "<f4"), ("n_ranges", "<i4"), ("ranges_offset", "<i8"), ("x", "<f8"), ("y", "<f8"),
                       ("yaw", "<f8"), ("x_end", "<f8"), ("y_end", "<f8"), ("yaw_end", "<f8")])
RANGE_READING_DTYPE = np.dtype([("range", "<f4"), ("max_range", "<f4"), ("x", "<f8"), ("y", "<f8"), ("yaw", "<f8")])
# the same records with the sensor's full pose (translation + quaternion x y z w): include/rna.h rna_laser_scan_tf / rna_range_reading_tf
SCAN_TF_DTYPE = np.dtype([("angle_min", "<f4"), ("angle_max", "<f4"), ("angle_increment", "<f4"), ("range_min", "<f4"),
                          ("range_max", "<f4"), ("n_ranges", "<i4"), ("ranges_offset", "<i8"), ("t", "<f8", (3,)), ("q", "<f8", (4,)),
                          ("t_end", "<f8", (3,)), ("q_end", "<f8", (4,))])
RANGE_READING_TF_DTYPE = np.dtype([("range", "<f4"), ("max_range", "<f4"), ("t", "<f8", (3,)), ("q", "<f8", (4,))])
ASTAR_QUERY_DTYPE = np.dtype([("start", "<i4"), ("goal", "<i4")])
ASTAR_RESULT_DTYPE = np.dtype([("status", "<i4"), ("path_len", "<i4"), ("cost", "<i4"), ("expanded", "<i4"),
                               ("rounds", "<i4"), ("buckets", "<i4")])
RRT_QUERY_DTYPE = np.dtype([("start", "<f8", (2,)), ("target", "<f8", (2,)), ("close_tolerance", "<f8"),
                            ("seed", "<u4"), ("max_samples", "<i4")])
RRT_RESULT_DTYPE = np.dtype([("status", "<i4"), ("path_len", "<i4"), ("tree_size", "<i4"), ("samples", "<i4")])

assert RAY_DTYPE.itemsize == 40 and POSE_DTYPE.itemsize == 48 and VFH_OUT_DTYPE.itemsize == 16
assert RRT_QUERY_DTYPE.itemsize == 48

_lib = None


class RnaError(RuntimeError):
    pass


def lib():
    """Loads librna.so; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RnaError("librna.so is not built: run `make -C ros_navigation_amd/csrc` (or __graft_entry__.build())")
    try:
        # torch bundles a libamdhip64 of its own (SONAME libamdhip64.so.7, the one librna.so asks for): loaded first,
        # both share ONE HIP runtime, so torch tensors and engine buffers live in the same context.  The other
        # order gives the process two runtimes and torch then finds no GPU.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.rna_abi_version.restype = C.c_int
    if L.rna_abi_version() != ABI_VERSION:
        raise RnaError("librna.so has ABI version %d, these bindings were written for %d: rebuild (make -C ros_navigation_amd/csrc)"
                       % (L.rna_abi_version(), ABI_VERSION))
    vp = C.c_void_p
    L.rna_create.argtypes = [C.POINTER(vp), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
    L.rna_destroy.argtypes = [vp]
    L.rna_destroy.restype = None
    L.rna_last_error.argtypes = [vp]
    L.rna_last_error.restype = C.c_char_p
    L.rna_get_geometry.argtypes = [vp, C.POINTER(Geometry)]
    L.rna_layer_upload.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.rna_layer_download.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.rna_layer_fill.argtypes = [vp, C.c_int, C.c_float]
    L.rna_layer_device_ptr.argtypes = [vp, C.c_int]
    L.rna_layer_device_ptr.restype = vp
    L.rna_stream.argtypes = [vp]
    L.rna_stream.restype = vp
    L.rna_synchronize.argtypes = [vp]
    L.rna_synchronize_map.argtypes = [vp]
    L.rna_hw_queue_advice.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.rna_get_index.argtypes = [vp, C.c_double, C.c_double, C.POINTER(C.c_int32)]
    L.rna_get_position.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    gp, ip = C.POINTER(Geometry), C.POINTER(C.c_int32)
    L.rna_geometry_index.argtypes = [gp, C.c_double, C.c_double, ip]
    L.rna_geometry_position.argtypes = [gp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.rna_line_cells.argtypes = [gp, C.c_double, C.c_double, C.c_double, C.c_double, ip, C.c_int]
    L.rna_circle_cells.argtypes = [gp, C.c_double, C.c_double, C.c_double, ip, C.c_int]
    L.rna_submap_cells.argtypes = [gp, ip, ip, ip, C.c_int]
    L.rna_clone.argtypes = [vp, C.POINTER(vp)]
    L.rna_scan_projected_beams.argtypes = [C.c_int, C.c_float]
    L.rna_himm_update.argtypes = [vp, C.c_int, vp, C.c_int]
    L.rna_himm_update_device.argtypes = [vp, C.c_int, vp, C.c_int]
    L.rna_compose_master.argtypes = [vp, C.c_int]
    L.rna_update_map.argtypes = [vp, vp, C.c_int, C.c_int]
    L.rna_update_map_device.argtypes = [vp, vp, C.c_int, C.c_int]
    L.rna_move.argtypes = [vp, C.c_double, C.c_double, C.POINTER(C.c_int)]
    L.rna_himm_set_window.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.rna_layer_pack_region.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_layer_unpack_region.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_layer_unpack_region_tracked.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_last_dirty_tiles.argtypes = [vp, vp, C.c_size_t]
    L.rna_layer_pack_tiles.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_layers_unpack_tiles.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_last_dirty_tiles_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.rna_layer_pack_tiles_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_layers_unpack_tiles_device.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rna_vfh_default_params.argtypes = [C.POINTER(VfhParams)]
    L.rna_vfh_default_params.restype = None
    L.rna_vfh_init.argtypes = [vp, C.POINTER(VfhParams), C.c_int]
    L.rna_vfh_reset.argtypes = [vp]
    L.rna_vfh_hist_size.argtypes = [vp]
    L.rna_vfh_step_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.rna_vfh_step_batch_device.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.rna_vfh_update_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    L.rna_astar_configure.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.rna_astar_set_pipeline_depth.argtypes = [vp, C.c_int]
    L.rna_astar_set_page_cap.argtypes = [vp, C.c_int]
    L.rna_astar_effective_config.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rna_astar_batch.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.rna_astar_batch_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.rna_astar_download_nbr_mask.argtypes = [vp, vp, C.c_size_t]
    L.rna_astar_settled_counts.argtypes = [vp, vp, C.c_int]
    if hasattr(L, "rna_astar_job_counters"):   # (absent only in an older build named by the developer switch RNA_LIB of bench.py's A/B runs)
        L.rna_astar_job_counters.argtypes = [vp, vp, C.c_int]
    L.rna_graph_astar_batch.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, vp]
    L.rna_rrt_batch.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.rna_rrt_batch_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.rna_to_occupancy_grid.argtypes = [vp, C.c_int, C.c_float, C.c_float, vp]
    L.rna_to_occupancy_grid_device.argtypes = [vp, C.c_int, C.c_float, C.c_float, vp]
    L.rna_from_occupancy_grid.argtypes = [vp, C.c_int, vp]
    L.rna_vfh_hist_msg_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.rna_tailor_plan.argtypes = [vp, C.c_int, C.c_uint, vp, C.POINTER(C.c_int)]
    L.rna_follow_plan.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_double, vp]
    L.rna_get_submap.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, vp, C.c_size_t,
                                 C.POINTER(SubmapInfo)]
    L.rna_get_submap_device.argtypes = L.rna_get_submap.argtypes
    L.rna_create_submap.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(vp)]
    L.rna_scan_to_rays.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, vp, C.c_int, C.POINTER(C.c_int)]
    L.rna_scan_to_rays_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]
    L.rna_range_to_rays.argtypes = [vp, C.c_int, vp]
    L.rna_scan_to_rays_tf.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, vp, C.c_int, C.POINTER(C.c_int)]
    L.rna_scan_to_rays_tf_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]
    L.rna_range_to_rays_tf.argtypes = [vp, C.c_int, vp]
    L.rna_profile_enable.argtypes = [vp, C.c_int]
    L.rna_profile_reset.argtypes = [vp]
    L.rna_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.rna_kernel_name.argtypes = [C.c_int]
    L.rna_kernel_name.restype = C.c_char_p
    _lib = L
    return L


def tailor_plan(plan_xy, stride=5):
    """Nav::taileredPlan: every stride-th position of the plan walked backwards, plus the last one."""
    plan = np.ascontiguousarray(plan_xy, np.float64).reshape(-1, 2)
    out = np.empty_like(plan)
    m = C.c_int(0)
    rc = lib().rna_tailor_plan(_ptr(plan), len(plan), stride, _ptr(out), C.byref(m))
    if rc != 0:
        raise RnaError("rna_tailor_plan failed (%d)" % rc)
    return out[:m.value].copy()


def range_to_rays(readings):
    """RangeMapUpdater::bufferIncomingMsg for a batch of sonar readings (RANGE_READING_DTYPE) -> RAY_DTYPE rays."""
    readings = np.ascontiguousarray(readings)
    assert readings.dtype == RANGE_READING_DTYPE
    rays = np.zeros(len(readings), RAY_DTYPE)
    rc = lib().rna_range_to_rays(_ptr(readings), len(readings), _ptr(rays))
    if rc != 0:
        raise RnaError("rna_range_to_rays failed (%d)" % rc)
    return rays


def range_to_rays_tf(readings):
    """rna_range_to_rays for sensors with a full pose (RANGE_READING_TF_DTYPE)."""
    readings = np.ascontiguousarray(readings)
    assert readings.dtype == RANGE_READING_TF_DTYPE
    rays = np.zeros(len(readings), RAY_DTYPE)
    rc = lib().rna_range_to_rays_tf(_ptr(readings), len(readings), _ptr(rays))
    if rc != 0:
        raise RnaError("rna_range_to_rays_tf failed (%d)" % rc)
    return rays


def follow_plan(plan_xy, plan_index, x, y, yaw, linear_velocity=0.0, dt=0.2):
    """Steerer::update's plan-following head: returns (following, plan_index, pose) -- pose is a POSE_DTYPE record
    ready for Engine.vfh_step, valid while following is True."""
    plan = np.ascontiguousarray(plan_xy, np.float64).reshape(-1, 2)
    idx = C.c_int32(plan_index)
    pose = np.zeros(1, POSE_DTYPE)
    rc = lib().rna_follow_plan(_ptr(plan), len(plan), C.byref(idx), x, y, yaw, linear_velocity, dt, _ptr(pose))
    if rc < 0:
        raise RnaError("rna_follow_plan failed (%d)" % rc)
    return rc == 1, idx.value, pose[0]


def default_vfh_params():
    p = VfhParams()
    lib().rna_vfh_default_params(C.byref(p))
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_geometry(length_x, length_y, resolution, position=(0.0, 0.0), start_index=(0, 0)):
    """rna_geometry as GridMap::setGeometry derives it (size = round(length / resolution), length = size * resolution)"""
    g = Geometry()
    g.size[0], g.size[1] = int(round(length_x / resolution)), int(round(length_y / resolution))
    g.resolution = resolution
    g.length[0], g.length[1] = g.size[0] * resolution, g.size[1] * resolution
    g.position[0], g.position[1] = position
    g.start_index[0], g.start_index[1] = start_index
    return g


def _cells(fn, *args):
    import numpy as _np
    cap = 4096
    while True:
        out = _np.zeros(2 * cap, _np.int32)
        n = fn(*args, out.ctypes.data_as(C.POINTER(C.c_int32)), cap)
        if n < 0:
            raise RnaError(STATUS.get(n, n))
        if n <= cap:
            return out[:2 * n].reshape(-1, 2).copy()
        cap = n


def line_cells(g, sx, sy, ex, ey):
    """LineIterator(map, start, end) as a cell list (host only; the cells a HIMM ray clears)"""
    return _cells(lib().rna_line_cells, C.byref(g), sx, sy, ex, ey)


def circle_cells(g, cx, cy, radius):
    return _cells(lib().rna_circle_cells, C.byref(g), cx, cy, radius)


def submap_cells(g, top_left, size):
    tl, sz = (C.c_int32 * 2)(*top_left), (C.c_int32 * 2)(*size)
    return _cells(lib().rna_submap_cells, C.byref(g), tl, sz)


def geometry_index(g, x, y):
    o = (C.c_int32 * 2)()
    rc = lib().rna_geometry_index(C.byref(g), x, y, o)
    return (o[0], o[1]) if rc == 1 else None


def geometry_position(g, i, j):
    o = (C.c_double * 2)()
    rc = lib().rna_geometry_position(C.byref(g), i, j, o)
    return (o[0], o[1]) if rc == 1 else None


class Engine:
    """One rna_engine: a device-resident GridMap (master/laser/range) plus the planners."""

    def __init__(self, length_x, length_y, resolution, pos_x=0.0, pos_y=0.0, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.rna_create(C.byref(h), length_x, length_y, resolution, pos_x, pos_y, device)
        if rc != RNA_OK:
            raise RnaError("rna_create failed: %s" % STATUS.get(rc, rc))
        self.h = h
        self.device = device
        g = self.geometry()
        self.rows, self.cols = g.size[0], g.size[1]
        self.ncell = self.rows * self.cols
        self.resolution = g.resolution
        self.hist_size = None

    def close(self):
        if getattr(self, "h", None):
            self._L.rna_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != RNA_OK:
            msg = self._L.rna_last_error(self.h)
            raise RnaError("%s: %s" % (STATUS.get(rc, rc), msg.decode() if msg else ""))

    # ---- container ----
    def geometry(self):
        g = Geometry()
        self._check(self._L.rna_get_geometry(self.h, C.byref(g)))
        return g

    def upload(self, layer, a):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        self._check(self._L.rna_layer_upload(self.h, layer, _ptr(a), a.size))

    def download(self, layer):
        a = np.empty(self.ncell, np.float32)
        self._check(self._L.rna_layer_download(self.h, layer, _ptr(a), a.size))
        return a

    def fill(self, layer, v):
        self._check(self._L.rna_layer_fill(self.h, layer, v))

    def layer_ptr(self, layer):
        return self._L.rna_layer_device_ptr(self.h, layer)

    def synchronize(self):
        self._check(self._L.rna_synchronize(self.h))

    def synchronize_map(self):
        """the map stream (+ the VFH+ side stream) only: A* batches in flight keep running"""
        self._check(self._L.rna_synchronize_map(self.h))

    def get_index(self, x, y):
        out = (C.c_int32 * 2)()
        ok = self._L.rna_get_index(self.h, x, y, out)
        return (out[0], out[1]) if ok == 1 else None

    def get_position(self, i, j):
        out = (C.c_double * 2)()
        ok = self._L.rna_get_position(self.h, i, j, out)
        return (out[0], out[1]) if ok == 1 else None

    def submap_engine(self, x, y, length_x, length_y):
        """GridMap::getSubmap as a GridMap of its own: a new Engine over the clamped window (all layers), or None."""
        h = C.c_void_p()
        rc = self._L.rna_create_submap(self.h, x, y, length_x, length_y, C.byref(h))
        if rc == 0:
            return None
        if rc != 1:
            self._check(rc)
        child = Engine.__new__(Engine)
        child._L, child.h, child.device = self._L, h, self.device
        g = child.geometry()
        child.rows, child.cols = g.size[0], g.size[1]
        child.ncell = child.rows * child.cols
        child.resolution = g.resolution
        child.hist_size = None
        return child

    def get_submap(self, layer, x, y, length_x, length_y):
        """GridMap::getSubmap: (info, data) with data column-major info.size[0] x info.size[1], or None when the
        reference's isSuccess is false."""
        info = SubmapInfo()
        cap = (int(np.ceil(length_x / self.resolution)) + 2) * (int(np.ceil(length_y / self.resolution)) + 2)
        cap = max(1, min(cap, self.ncell))
        out = np.empty(cap, np.float32)
        rc = self._L.rna_get_submap(self.h, layer, x, y, length_x, length_y, _ptr(out), out.size, C.byref(info))
        if rc == 0:
            return None
        if rc != 1:
            self._check(rc)
        return info, out[:info.size[0] * info.size[1]].copy()

    def move(self, x, y):
        m = C.c_int(0)
        self._check(self._L.rna_move(self.h, x, y, C.byref(m)))
        return bool(m.value)

    # ---- HIMM ----
    def himm_update(self, layer, rays):
        assert rays.dtype == RAY_DTYPE
        rays = np.ascontiguousarray(rays)
        self._check(self._L.rna_himm_update(self.h, layer, _ptr(rays), len(rays)))

    def himm_update_device(self, layer, rays_ptr, n):
        self._check(self._L.rna_himm_update_device(self.h, layer, rays_ptr, n))

    def compose_master(self, mode=0):
        self._check(self._L.rna_compose_master(self.h, mode))

    def himm_set_window(self, i0=0, j0=0, ni=0, nj=0):
        """Owner window of the tiled single-map mode (ni == 0: whole map)."""
        self._check(self._L.rna_himm_set_window(self.h, i0, j0, ni, nj))

    def pack_region(self, layer, i0, ni, j0, nj):
        """Block [i0,i0+ni) x [j0,j0+nj) of a layer as a dense device tensor of ni*nj floats (i fastest)."""
        import torch
        t = torch.empty(max(ni, 0) * max(nj, 0), dtype=torch.float32, device=self._torch_device())
        if t.numel():
            self._check(self._L.rna_layer_pack_region(self.h, layer, i0, ni, j0, nj, t.data_ptr()))
        return t

    def unpack_region(self, layer, i0, ni, j0, nj, t, tracked=False):
        """tracked: flag the covered 64 x 64 tiles for the next compose_master(0) instead of "whole layer changed"."""
        if ni <= 0 or nj <= 0:
            return
        assert t.is_cuda and t.dtype.is_floating_point and t.numel() >= ni * nj and t.is_contiguous()
        import torch
        torch.cuda.current_stream(t.device).synchronize()   # the tensor was filled on torch's stream
        fn = self._L.rna_layer_unpack_region_tracked if tracked else self._L.rna_layer_unpack_region
        self._check(fn(self.h, layer, i0, ni, j0, nj, t.data_ptr()))

    TILE = 64

    def tile_grid(self):
        return (self.rows + self.TILE - 1) // self.TILE, (self.cols + self.TILE - 1) // self.TILE

    def last_dirty_tiles(self):
        """uint8 flag per 64 x 64 tile (index tj * tiles_i + ti): what the last compose_master consumed."""
        ti, tj = self.tile_grid()
        f = np.zeros(ti * tj, np.uint8)
        self._check(self._L.rna_last_dirty_tiles(self.h, _ptr(f), f.size))
        return f

    def pack_tiles(self, layer, tiles, window):
        """The listed tiles, clipped to window = (i0, ni, j0, nj), as a dense device tensor of len(tiles) x 4096 floats."""
        import torch
        tiles = np.ascontiguousarray(tiles, np.int32)
        t = torch.empty(len(tiles) * self.TILE * self.TILE, dtype=torch.float32, device=self._torch_device())
        if len(tiles):
            i0, ni, j0, nj = window
            self._check(self._L.rna_layer_pack_tiles(self.h, layer, _ptr(tiles), len(tiles), i0, ni, j0, nj, t.data_ptr()))
        return t

    def unpack_tiles(self, layers, tiles, window, t):
        """Inverse of pack_tiles into one or two layers (e.g. (LASER, MASTER)); flags the tiles for compose_master(0)."""
        tiles = np.ascontiguousarray(tiles, np.int32)
        if not len(tiles):
            return
        import torch
        assert t.is_cuda and t.is_contiguous() and t.numel() >= len(tiles) * self.TILE * self.TILE
        torch.cuda.current_stream(t.device).synchronize()
        la, lb = (layers[0], layers[1]) if len(layers) > 1 else (layers[0], -1)
        i0, ni, j0, nj = window
        self._check(self._L.rna_layers_unpack_tiles(self.h, la, lb, _ptr(tiles), len(tiles), i0, ni, j0, nj, t.data_ptr()))

    def last_dirty_tiles_device(self, window, list_ptr, count_ptr):
        """the flagged tiles that intersect window = (i0, ni, j0, nj), compacted into a device list (room for every tile
        of the map) and their number into a device int; asynchronous on the engine's stream"""
        i0, ni, j0, nj = window
        self._check(self._L.rna_last_dirty_tiles_device(self.h, i0, ni, j0, nj, list_ptr, count_ptr))

    def pack_tiles_device(self, layer, list_ptr, n, window, dense_ptr):
        i0, ni, j0, nj = window
        if n:
            self._check(self._L.rna_layer_pack_tiles_device(self.h, layer, list_ptr, n, i0, ni, j0, nj, dense_ptr))

    def unpack_tiles_device(self, layers, list_ptr, n, window, dense_ptr):
        la, lb = (layers[0], layers[1]) if len(layers) > 1 else (layers[0], -1)
        i0, ni, j0, nj = window
        if n:
            self._check(self._L.rna_layers_unpack_tiles_device(self.h, la, lb, list_ptr, n, i0, ni, j0, nj, dense_ptr))

    def _torch_device(self):
        import torch
        return torch.device("cuda", self.device)

    def update_map(self, rays, compose_mode=0):
        assert rays.dtype == RAY_DTYPE
        rays = np.ascontiguousarray(rays)
        self._check(self._L.rna_update_map(self.h, _ptr(rays), len(rays), compose_mode))

    def update_map_device(self, rays_ptr, n, compose_mode=0):
        self._check(self._L.rna_update_map_device(self.h, rays_ptr, n, compose_mode))

    # ---- VFH ----
    def vfh_init(self, n_robots, params=None):
        p = params or default_vfh_params()
        self._check(self._L.rna_vfh_init(self.h, C.byref(p), n_robots))
        self.hist_size = self._L.rna_vfh_hist_size(self.h)
        self.n_robots = n_robots

    def vfh_reset(self):
        self._check(self._L.rna_vfh_reset(self.h))

    def vfh_step(self, poses, want_hist=True):
        assert poses.dtype == POSE_DTYPE
        poses = np.ascontiguousarray(poses)
        n = len(poses)
        out = np.zeros(n, VFH_OUT_DTYPE)
        origin = np.zeros((n, self.hist_size), np.float32) if want_hist else None
        hist = np.zeros((n, self.hist_size), np.float32) if want_hist else None
        self._check(self._L.rna_vfh_step_batch(self.h, _ptr(poses), n, _ptr(out),
                                               _ptr(origin) if want_hist else None,
                                               _ptr(hist) if want_hist else None))
        return out, origin, hist

    def vfh_step_device(self, poses_ptr, n, out_ptr, origin_ptr=None, hist_ptr=None):
        self._check(self._L.rna_vfh_step_batch_device(self.h, poses_ptr, n, out_ptr, origin_ptr, hist_ptr))

    def vfh_update(self, ranges, poses):
        """VFH::Update_VFH on caller-provided scans: ranges is (n, 361, 2) float64."""
        assert poses.dtype == POSE_DTYPE
        ranges = np.ascontiguousarray(ranges, dtype=np.float64)
        n = len(poses)
        assert ranges.shape == (n, 361, 2)
        poses = np.ascontiguousarray(poses)
        out = np.zeros(n, VFH_OUT_DTYPE)
        origin = np.zeros((n, self.hist_size), np.float32)
        hist = np.zeros((n, self.hist_size), np.float32)
        self._check(self._L.rna_vfh_update_batch(self.h, _ptr(ranges), _ptr(poses), n, _ptr(out), _ptr(origin),
                                                 _ptr(hist)))
        return out, origin, hist

    # ---- planners ----
    def astar_configure(self, max_queries=0, queue_capacity=0, bucket_width=0):
        self._check(self._L.rna_astar_configure(self.h, max_queries, queue_capacity, bucket_width))

    def astar_pipeline_depth(self, depth):
        self._check(self._L.rna_astar_set_pipeline_depth(self.h, depth))

    def astar_effective_config(self):
        """(pipeline depth, pages per query, concurrent queries) as allocated; zeros before the first batch"""
        d, p, q = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._L.rna_astar_effective_config(self.h, C.byref(d), C.byref(p), C.byref(q)))
        return d.value, p.value, q.value

    def astar_page_cap(self, pages_per_query):
        self._check(self._L.rna_astar_set_page_cap(self.h, pages_per_query))

    def astar(self, queries, max_path_len):
        assert queries.dtype == ASTAR_QUERY_DTYPE
        queries = np.ascontiguousarray(queries)
        n = len(queries)
        paths = np.zeros((n, max_path_len), np.int32)
        res = np.zeros(n, ASTAR_RESULT_DTYPE)
        self._check(self._L.rna_astar_batch(self.h, _ptr(queries), n, _ptr(paths), max_path_len, _ptr(res)))
        return res, paths

    def astar_device(self, queries_ptr, n, paths_ptr, max_path_len, results_ptr):
        self._check(self._L.rna_astar_batch_device(self.h, queries_ptr, n, paths_ptr, max_path_len, results_ptr))

    def astar_settled(self, n):
        """E per query of the last (single-chunk) batch: |{cells : g + h <= f*}|."""
        out = np.zeros(n, np.int32)
        self._check(self._L.rna_astar_settled_counts(self.h, _ptr(out), n))
        return out

    def astar_job_counters(self, reset=False):
        """What the search kernels counted since the last reset (rna_astar_job_counters): a dict of totals over all searches."""
        out = np.zeros(8, np.uint64)
        if not hasattr(self._L, "rna_astar_job_counters"):
            return None
        self._check(self._L.rna_astar_job_counters(self.h, _ptr(out), 1 if reset else 0))
        return dict(zip(("searches", "tiles_touched", "jobs", "jobs_noop", "sticky_turns", "rows_written", "buckets", "bucket_reruns"), (int(v) for v in out)))

    def nbr_mask(self):
        a = np.empty(self.ncell, np.uint8)
        self._check(self._L.rna_astar_download_nbr_mask(self.h, _ptr(a), a.size))
        return a

    def graph_astar(self, vertex_xy, edge_uv, start_target, edge_weight=None, max_len=None):
        v = np.ascontiguousarray(vertex_xy, dtype=np.float64).reshape(-1, 2)
        e = np.ascontiguousarray(edge_uv, dtype=np.int32).reshape(-1, 2)
        st = np.ascontiguousarray(start_target, dtype=np.float64).reshape(-1, 4)
        w = None if edge_weight is None else np.ascontiguousarray(edge_weight, dtype=np.float32)
        n = len(st)
        max_len = max_len or (len(v) + 2)
        paths = np.zeros((n, max_len, 2), np.float64)
        plen = np.zeros(n, np.int32)
        self._check(self._L.rna_graph_astar_batch(self.h, len(v), _ptr(v), len(e), _ptr(e),
                                                  _ptr(w) if w is not None else None, _ptr(st), n, _ptr(paths),
                                                  max_len, _ptr(plen)))
        return plen, paths

    def rrt(self, queries, max_path_len=2048):
        assert queries.dtype == RRT_QUERY_DTYPE
        queries = np.ascontiguousarray(queries)
        n = len(queries)
        paths = np.zeros((n, max_path_len, 2), np.float64)
        res = np.zeros(n, RRT_RESULT_DTYPE)
        self._check(self._L.rna_rrt_batch(self.h, _ptr(queries), n, _ptr(paths), max_path_len, _ptr(res)))
        return res, paths

    def rrt_device(self, queries_ptr, n, paths_ptr, max_path_len, results_ptr):
        self._check(self._L.rna_rrt_batch_device(self.h, queries_ptr, n, paths_ptr, max_path_len, results_ptr))

    # ---- measurement ----
    def scan_to_rays(self, scans, ranges, max_rays=None):
        """LaserMapUpdater::bufferIncomingMsg for a batch of scans: RAY_DTYPE array in scan order, beam order."""
        scans = np.ascontiguousarray(scans, SCAN_DTYPE)
        ranges = np.ascontiguousarray(ranges, np.float32)
        if max_rays is None:
            max_rays = int(scans["n_ranges"].sum()) + 1
        rays = np.zeros(max_rays, RAY_DTYPE)
        n = C.c_int(0)
        self._check(self._L.rna_scan_to_rays(self.h, _ptr(scans), len(scans), _ptr(ranges), ranges.size, _ptr(rays), max_rays,
                                             C.byref(n)))
        return rays[:n.value]

    def scan_to_rays_tf(self, scans, ranges, max_rays=None):
        """scan_to_rays for sensors with a full pose (SCAN_TF_DTYPE: tf's translation + quaternion at both ends of the scan)."""
        scans = np.ascontiguousarray(scans, SCAN_TF_DTYPE)
        ranges = np.ascontiguousarray(ranges, np.float32)
        if max_rays is None:
            max_rays = int(scans["n_ranges"].sum()) + 1
        rays = np.zeros(max_rays, RAY_DTYPE)
        n = C.c_int(0)
        self._check(self._L.rna_scan_to_rays_tf(self.h, _ptr(scans), len(scans), _ptr(ranges), ranges.size, _ptr(rays), max_rays,
                                                C.byref(n)))
        return rays[:n.value]

    def to_occupancy_grid(self, layer, data_min=0.0, data_max=255.0):
        """GridMapRosConverter::toOccupancyGrid: int8[rows*cols] in nav_msgs/OccupancyGrid order."""
        g = self.geometry()
        out = np.empty(g.size[0] * g.size[1], np.int8)
        self._check(self._L.rna_to_occupancy_grid(self.h, layer, data_min, data_max, _ptr(out)))
        return out

    def to_occupancy_grid_device(self, layer, data_min, data_max, out_ptr):
        self._check(self._L.rna_to_occupancy_grid_device(self.h, layer, data_min, data_max, out_ptr))

    def from_occupancy_grid(self, layer, data):
        data = np.ascontiguousarray(data, np.int8)
        self._check(self._L.rna_from_occupancy_grid(self.h, layer, _ptr(data)))

    def vfh_hist_msg(self, n):
        """Steerer::pubHist for the first n robots: (xData[bins], yData[n, bins], yBinData[n, bins], (low, high))."""
        bins = self._L.rna_vfh_hist_size(self.h) // 2
        x = np.empty(bins, np.uint16)
        y = np.empty((n, bins), np.uint16)
        yb = np.empty((n, bins), np.uint16)
        th = np.empty(2, np.uint16)
        self._check(self._L.rna_vfh_hist_msg_batch(self.h, n, _ptr(x), _ptr(y), _ptr(yb), _ptr(th)))
        return x, y, yb, (int(th[0]), int(th[1]))

    def profile(self, on=True):
        """True / 1: every kernel slot, 2: only the slots on the A* stages' own streams, False / 0: off"""
        self._check(self._L.rna_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._check(self._L.rna_profile_reset(self.h))

    def profile_get(self):
        out = {}
        for i, name in enumerate(KERNELS):
            ms, cnt = C.c_double(0), C.c_int64(0)
            self._check(self._L.rna_profile_get(self.h, i, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out
