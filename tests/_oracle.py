"""ctypes access to the CPU oracle (oracle/librna_oracle.so) and, when built, to the reference's own
vfh.cpp (oracle/_ref/libref_vfh.so).  TEST INFRASTRUCTURE ONLY -- the product package never imports
this module.  Built on demand with `make -C oracle`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("RNA_ORACLE_SO") or os.path.join(ORACLE_DIR, "librna_oracle.so")   # override: sanitizer build
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_vfh.so")


class Geom(C.Structure):
    _fields_ = [("len", C.c_double * 2), ("pos", C.c_double * 2), ("res", C.c_double),
                ("size", C.c_int * 2), ("start", C.c_int * 2)]


class SubmapInfo(C.Structure):
    _fields_ = [("top_left", C.c_int * 2), ("size", C.c_int * 2), ("pos", C.c_double * 2),
                ("len", C.c_double * 2), ("requested_index", C.c_int * 2)]


class Region(C.Structure):
    _fields_ = [("index", C.c_int * 2), ("size", C.c_int * 2), ("quadrant", C.c_int)]


class Ray(C.Structure):
    _fields_ = [("sx", C.c_double), ("sy", C.c_double), ("ex", C.c_double), ("ey", C.c_double),
                ("clear_end", C.c_int32), ("_pad", C.c_int32)]


RAY_DTYPE = np.dtype([("sx", "<f8"), ("sy", "<f8"), ("ex", "<f8"), ("ey", "<f8"),
                      ("clear_end", "<i4"), ("_pad", "<i4")])


class VfhParams(C.Structure):
    _fields_ = [("cell_size", C.c_double), ("window_diameter", C.c_int), ("sector_angle", C.c_int),
                ("safety_dist_0ms", C.c_double), ("safety_dist_1ms", C.c_double),
                ("max_speed", C.c_int), ("max_speed_narrow_opening", C.c_int),
                ("max_speed_wide_opening", C.c_int), ("max_acceleration", C.c_int),
                ("min_turnrate", C.c_int), ("max_turnrate_0ms", C.c_int), ("max_turnrate_1ms", C.c_int),
                ("min_turn_radius_safety_factor", C.c_double),
                ("free_space_cutoff_0ms", C.c_double), ("obs_cutoff_0ms", C.c_double),
                ("free_space_cutoff_1ms", C.c_double), ("obs_cutoff_1ms", C.c_double),
                ("weight_desired_dir", C.c_double), ("weight_current_dir", C.c_double),
                ("robot_radius", C.c_double)]


class AstarResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("path_len", C.c_int32), ("cost", C.c_int32), ("settled", C.c_int32)]


class RrtResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("path_len", C.c_int32), ("tree_size", C.c_int32), ("samples", C.c_int32)]


class RandState(C.Structure):
    _fields_ = [("r", C.c_int32 * 34), ("f", C.c_int), ("b", C.c_int)]


Ranges = (C.c_double * 2) * 361
_lib = None
_ref = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        d2 = C.POINTER(C.c_double)
        i2 = C.POINTER(C.c_int)
        fp = C.POINTER(C.c_float)
        L.og_set_geometry.argtypes = [C.POINTER(Geom)] + [C.c_double] * 5
        L.og_position_from_index.argtypes = [C.POINTER(Geom), i2, d2]
        L.og_index_from_position.argtypes = [C.POINTER(Geom), d2, i2]
        L.og_position_within_map.argtypes = [d2, d2, d2]
        L.og_index_shift_from_position_shift.argtypes = [d2, C.c_double, i2]
        L.og_position_shift_from_index_shift.argtypes = [i2, C.c_double, d2]
        L.og_index_within_range.argtypes = [i2, i2]
        L.og_wrap_index.argtypes = [C.c_int, C.c_int]
        L.og_limit_position_to_range.argtypes = [d2, d2, d2]
        L.og_increment_index.argtypes = [i2, i2, i2]
        L.og_increment_index_for_submap.argtypes = [i2, i2, i2, i2, i2, i2]
        L.og_index_from_linear.argtypes = [C.c_size_t, i2, C.c_int, i2]
        L.og_submap_information.argtypes = [C.POINTER(Geom), d2, d2, C.POINTER(SubmapInfo)]
        L.og_buffer_regions_for_submap.argtypes = [i2, i2, i2, i2, C.POINTER(Region)]
        L.og_get_submap.argtypes = [C.POINTER(Geom), fp, d2, d2, C.POINTER(Geom), fp, C.c_int]
        L.og_move.argtypes = [C.POINTER(Geom), C.POINTER(fp), C.c_int, d2, C.POINTER(Region), i2]
        L.og_line_cells.argtypes = [C.POINTER(Geom), d2, d2, i2, C.c_int]
        L.og_line_cells_index.argtypes = [i2, i2, i2, C.c_int]
        L.og_circle_cells.argtypes = [C.POINTER(Geom), d2, C.c_double, i2, C.c_int]
        L.og_submap_cells.argtypes = [C.POINTER(Geom), i2, i2, i2, C.c_int]
        L.og_himm_clear.argtypes = [C.c_float]
        L.og_himm_clear.restype = C.c_float
        L.og_himm_mark.argtypes = [C.c_float]
        L.og_himm_mark.restype = C.c_float
        L.og_himm_update.argtypes = [C.POINTER(Geom), fp, C.c_void_p, C.c_int, d2]
        L.og_himm_update.restype = None
        L.og_ranges_from_submap.argtypes = [C.POINTER(Geom), fp, d2, C.c_double, C.POINTER(Ranges)]
        L.og_vfh_default_params.argtypes = [C.POINTER(VfhParams)]
        L.og_vfh_create.argtypes = [C.POINTER(VfhParams)]
        L.og_vfh_create.restype = C.c_void_p
        L.og_vfh_destroy.argtypes = [C.c_void_p]
        L.og_vfh_update.argtypes = [C.c_void_p, C.POINTER(Ranges), C.c_int, C.c_float, C.c_float, C.c_float,
                                    C.c_double, i2, i2]
        L.og_vfh_step_pose.argtypes = [C.c_void_p, C.POINTER(Geom), fp, d2, C.c_double, C.c_int, C.c_float,
                                       C.c_float, C.c_float, C.c_double, i2, i2]
        L.og_vfh_hist_size.argtypes = [C.c_void_p]
        for n in ("og_vfh_hist", "og_vfh_origin_hist", "og_vfh_cell_direction", "og_vfh_cell_dist",
                  "og_vfh_cell_base_mag"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = fp
        L.og_vfh_picked_angle.argtypes = [C.c_void_p]
        L.og_vfh_picked_angle.restype = C.c_float
        L.og_vfh_last_picked_angle.argtypes = [C.c_void_p]
        L.og_vfh_last_picked_angle.restype = C.c_float
        L.og_vfh_max_speed_for_picked_angle.argtypes = [C.c_void_p]
        L.og_vfh_num_tables.argtypes = [C.c_void_p]
        L.og_vfh_cell_sector_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.og_vfh_cell_sector_list.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.og_vfh_cell_sector_list.restype = i2
        L.og_vfh_min_turning_radius.argtypes = [C.c_void_p, C.c_int]
        L.og_astar_blocked_mask.argtypes = [fp, C.c_size_t, C.POINTER(C.c_uint8)]
        L.og_astar_nbr_mask.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.og_astar_query.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(AstarResult)]
        L.og_astar_query.restype = None
        L.og_astar_query_on_map.argtypes = [C.POINTER(Geom), fp, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.c_int, C.POINTER(AstarResult)]
        L.og_astar_query_on_map.restype = None
        L.og_graph_astar.argtypes = [C.c_int, d2, C.c_int, i2, fp, C.c_int, C.c_int, i2, C.c_int]
        L.og_graph_closest_vertex.argtypes = [C.c_int, d2, d2]
        L.og_reference_graph.argtypes = [d2, i2]
        L.og_graph_make_plan.argtypes = [d2, d2, d2, C.c_int]
        L.og_srand.argtypes = [C.POINTER(RandState), C.c_uint]
        L.og_rand.argtypes = [C.POINTER(RandState)]
        L.og_if_blocked.argtypes = [C.POINTER(Geom), fp, d2]
        L.og_rrt_plan.argtypes = [C.POINTER(Geom), fp, d2, d2, C.c_double, C.c_uint, C.c_int, d2, C.c_int,
                                  C.POINTER(RrtResult)]
        L.og_rrt_plan.restype = None
        L.og_rrt_plan_steer.argtypes = [C.POINTER(Geom), fp, d2, d2, C.c_double, C.c_uint, C.c_int, C.c_int, d2, C.c_int,
                                        C.POINTER(RrtResult)]
        L.og_rrt_plan_steer.restype = None
        L.og_scan_to_rays.argtypes = [C.c_void_p, fp, C.c_void_p, C.c_int]
        L.og_scan_to_rays_tf.argtypes = [C.c_void_p, fp, C.c_void_p, C.c_int]
        L.og_range_to_ray_tf.argtypes = [C.c_float, C.c_float, d2, d2, C.c_void_p]
        L.og_range_to_ray_tf.restype = None
        L.og_simplify_scan.argtypes = [C.c_int, C.c_float, i2, C.c_int, fp]
        L.og_to_occupancy_grid.argtypes = [C.POINTER(Geom), fp, C.c_float, C.c_float, C.c_void_p]
        L.og_to_occupancy_grid.restype = None
        L.og_from_occupancy_grid.argtypes = [C.c_int, C.c_int, C.c_void_p, fp]
        L.og_from_occupancy_grid.restype = None
        L.og_hist_msg.argtypes = [fp, fp, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.og_tailor_plan.argtypes = [d2, C.c_int, C.c_uint, d2]
        L.og_range_to_ray.argtypes = [C.c_float, C.c_float, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.og_range_to_ray.restype = None
        L.og_follow_plan.argtypes = [d2, C.c_int, C.POINTER(C.c_int), C.c_double, C.c_double, C.c_double,
                                     C.POINTER(C.c_float)]
        _lib = L
    return _lib


def ref():
    """The reference's own vfh.cpp (None when oracle/_ref was never built)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            return None
        R = C.CDLL(REF_SO)
        R.refvfh_set_clock.argtypes = [C.c_double]
        R.refvfh_create.argtypes = [C.POINTER(VfhParams)]
        R.refvfh_create.restype = C.c_void_p
        R.refvfh_destroy.argtypes = [C.c_void_p]
        R.refvfh_update.argtypes = [C.c_void_p, C.POINTER(Ranges), C.c_int, C.c_float, C.c_float, C.c_float,
                                    C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        R.refvfh_hist_size.argtypes = [C.c_void_p]
        R.refvfh_hist.argtypes = [C.c_void_p]
        R.refvfh_hist.restype = C.POINTER(C.c_float)
        R.refvfh_origin_hist.argtypes = [C.c_void_p]
        R.refvfh_origin_hist.restype = C.POINTER(C.c_float)
        R.refvfh_picked_angle.argtypes = [C.c_void_p]
        R.refvfh_picked_angle.restype = C.c_float
        R.refvfh_num_tables.argtypes = [C.c_void_p]
        for n in ("refvfh_cell_direction", "refvfh_cell_dist", "refvfh_cell_base_mag"):
            getattr(R, n).argtypes = [C.c_void_p, C.c_int, C.c_int]
            getattr(R, n).restype = C.c_float
        R.refvfh_cell_sector_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        R.refvfh_cell_sector.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        R.refvfh_min_turning_radius.argtypes = [C.c_void_p, C.c_int]
        _ref = R
    return _ref


# ---------------------------------------------------------------------------------------------
# numpy-friendly helpers
# ---------------------------------------------------------------------------------------------
def d2(x, y):
    return (C.c_double * 2)(x, y)


def i2(a, b):
    return (C.c_int * 2)(a, b)


def make_geom(len_x, len_y, res, px=0.0, py=0.0, start=(0, 0)):
    g = Geom()
    lib().og_set_geometry(C.byref(g), len_x, len_y, res, px, py)
    g.start[0], g.start[1] = start
    return g


def raw_geom(length, pos, res, size, start=(0, 0)):
    g = Geom()
    g.len[0], g.len[1] = length
    g.pos[0], g.pos[1] = pos
    g.res = res
    g.size[0], g.size[1] = size
    g.start[0], g.start[1] = start
    return g


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_vfh_params():
    p = VfhParams()
    lib().og_vfh_default_params(C.byref(p))
    return p


def himm_update(g, layer, rays):
    """layer: float32 1-D array of rows*cols in column-major order (modified in place)."""
    assert rays.dtype == RAY_DTYPE
    lib().og_himm_update(C.byref(g), fptr(layer), rays.ctypes.data, len(rays), None)


def ranges_from_submap(g, master, x, y, yaw):
    r = Ranges()
    ok = lib().og_ranges_from_submap(C.byref(g), fptr(master), d2(x, y), yaw, C.byref(r))
    out = np.frombuffer(r, dtype=np.float64).reshape(361, 2)[:, 0].copy()
    return ok, out


def ranges_array(vals):
    r = Ranges()
    for i in range(361):
        r[i][0] = float(vals[i])
        r[i][1] = 0.0
    return r


class OracleVfh:
    def __init__(self, params=None):
        self.p = params or default_vfh_params()
        self.h = lib().og_vfh_create(C.byref(self.p))
        self.H = lib().og_vfh_hist_size(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().og_vfh_destroy(self.h)
            self.h = None

    def update(self, ranges, speed, gdir, gdist, gtol, dt):
        cs, ct = C.c_int(0), C.c_int(0)
        r = ranges if isinstance(ranges, Ranges) else ranges_array(ranges)
        lib().og_vfh_update(self.h, C.byref(r), speed, gdir, gdist, gtol, dt, C.byref(cs), C.byref(ct))
        return cs.value, ct.value

    def step_pose(self, g, master, x, y, yaw, speed, gdir, gdist, gtol, dt):
        cs, ct = C.c_int(0), C.c_int(0)
        lib().og_vfh_step_pose(self.h, C.byref(g), fptr(master), d2(x, y), yaw, speed, gdir, gdist, gtol, dt,
                               C.byref(cs), C.byref(ct))
        return cs.value, ct.value

    def hist(self):
        return np.ctypeslib.as_array(lib().og_vfh_hist(self.h), (self.H,)).copy()

    def origin_hist(self):
        return np.ctypeslib.as_array(lib().og_vfh_origin_hist(self.h), (self.H,)).copy()

    def picked_angle(self):
        return lib().og_vfh_picked_angle(self.h)


class RefVfh:
    """The reference's VFH class itself (oracle/_ref)."""

    def __init__(self, params=None):
        self.p = params or default_vfh_params()
        self.h = ref().refvfh_create(C.byref(self.p))
        self.H = ref().refvfh_hist_size(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            ref().refvfh_destroy(self.h)
            self.h = None

    def update(self, ranges, speed, gdir, gdist, gtol, dt):
        cs, ct = C.c_int(0), C.c_int(0)
        r = ranges if isinstance(ranges, Ranges) else ranges_array(ranges)
        ref().refvfh_update(self.h, C.byref(r), speed, gdir, gdist, gtol, dt, C.byref(cs), C.byref(ct))
        return cs.value, ct.value

    def hist(self):
        return np.ctypeslib.as_array(ref().refvfh_hist(self.h), (self.H,)).copy()

    def origin_hist(self):
        return np.ctypeslib.as_array(ref().refvfh_origin_hist(self.h), (self.H,)).copy()

    def picked_angle(self):
        return ref().refvfh_picked_angle(self.h)


def astar_masks(master, rows, cols):
    blocked = np.zeros(rows * cols, np.uint8)
    nbr = np.zeros(rows * cols, np.uint8)
    u8 = C.POINTER(C.c_uint8)
    lib().og_astar_blocked_mask(fptr(master), master.size, blocked.ctypes.data_as(u8))
    lib().og_astar_nbr_mask(blocked.ctypes.data_as(u8), rows, cols, nbr.ctypes.data_as(u8))
    return blocked, nbr


def astar_query(nbr, rows, cols, start, goal, path_cap=None, g_work=None):
    path_cap = path_cap or rows * cols
    if g_work is None:
        g_work = np.empty(rows * cols, np.int32)
    path = np.empty(path_cap, np.int32)
    res = AstarResult()
    lib().og_astar_query(nbr.ctypes.data_as(C.POINTER(C.c_uint8)), rows, cols, int(start), int(goal),
                         g_work.ctypes.data_as(C.POINTER(C.c_int32)), path.ctypes.data_as(C.POINTER(C.c_int32)),
                         path_cap, C.byref(res))
    n = res.path_len if res.status == 0 else 0
    return res, path[:n].copy(), g_work


def astar_last_settled_at_goal():
    """cells the calling thread's last astar_query had closed when the goal came off the heap (stop-at-goal A*)"""
    f = lib().og_astar_last_settled_at_goal
    f.restype = C.c_int32
    return int(f())


def astar_query_on_map(g, master, start, goal, path_cap=None):
    """grid A* in map space on a (possibly moved) map; start/goal/path are buffer linear indices"""
    n = g.size[0] * g.size[1]
    path_cap = path_cap or n
    g_work = np.empty(n, np.int32)
    path = np.empty(path_cap, np.int32)
    res = AstarResult()
    lib().og_astar_query_on_map(C.byref(g), fptr(master), int(start), int(goal), g_work.ctypes.data_as(C.POINTER(C.c_int32)),
                                path.ctypes.data_as(C.POINTER(C.c_int32)), path_cap, C.byref(res))
    k = res.path_len if res.status == 0 else 0
    return res, path[:k].copy()


def rrt_plan(g, master, start, target, tol=0.2, seed=1, max_samples=200000, cap=2048, steer=0):
    """steer: 0 = the reference's atan2 / cos / sin (glibc), 1 = the normalised-offset form the HIP kernel uses (oracle/rrt.c)"""
    path = np.zeros(2 * cap, np.float64)
    res = RrtResult()
    lib().og_rrt_plan_steer(C.byref(g), fptr(master), d2(*start), d2(*target), tol, seed, max_samples, steer,
                            path.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(res))
    return res, path[:2 * min(res.path_len, cap)].reshape(-1, 2).copy()


SCAN_DTYPE = np.dtype([("angle_min", "<f4"), ("angle_max", "<f4"), ("angle_increment", "<f4"), ("range_min", "<f4"),
                       ("range_max", "<f4"), ("n_ranges", "<i4"), ("ranges_offset", "<i8"), ("x", "<f8"), ("y", "<f8"),
                       ("yaw", "<f8"), ("x_end", "<f8"), ("y_end", "<f8"), ("yaw_end", "<f8")])


def scan_to_rays(scans, ranges):
    """LaserMapUpdater::bufferIncomingMsg for every scan, concatenated in order"""
    scans = np.ascontiguousarray(scans, SCAN_DTYPE)
    ranges = np.ascontiguousarray(ranges, np.float32)
    out = []
    for k in range(len(scans)):
        cap = int(scans["n_ranges"][k]) + 1
        rays = np.zeros(cap, RAY_DTYPE)
        n = lib().og_scan_to_rays(scans[k:k + 1].ctypes.data, fptr(ranges), rays.ctypes.data, cap)
        assert n <= cap
        out.append(rays[:n])
    return np.concatenate(out) if out else np.zeros(0, RAY_DTYPE)


SCAN_TF_DTYPE = np.dtype([("angle_min", "<f4"), ("angle_max", "<f4"), ("angle_increment", "<f4"), ("range_min", "<f4"),
                          ("range_max", "<f4"), ("n_ranges", "<i4"), ("ranges_offset", "<i8"), ("t", "<f8", (3,)), ("q", "<f8", (4,)),
                          ("t_end", "<f8", (3,)), ("q_end", "<f8", (4,))])


def scan_to_rays_tf(scans, ranges):
    """the same for sensors with a full pose (og_scan_tf: tf's translation + quaternion at both ends of the scan)"""
    scans = np.ascontiguousarray(scans, SCAN_TF_DTYPE)
    ranges = np.ascontiguousarray(ranges, np.float32)
    out = []
    for k in range(len(scans)):
        cap = int(scans["n_ranges"][k]) + 1
        rays = np.zeros(cap, RAY_DTYPE)
        n = lib().og_scan_to_rays_tf(scans[k:k + 1].ctypes.data, fptr(ranges), rays.ctypes.data, cap)
        assert n <= cap
        out.append(rays[:n])
    return np.concatenate(out) if out else np.zeros(0, RAY_DTYPE)


def range_to_rays_tf(readings):
    rays = np.zeros(len(readings), RAY_DTYPE)
    for k, m in enumerate(readings):
        t = np.ascontiguousarray(m["t"], np.float64)
        q = np.ascontiguousarray(m["q"], np.float64)
        d2 = C.POINTER(C.c_double)
        lib().og_range_to_ray_tf(float(m["range"]), float(m["max_range"]), t.ctypes.data_as(d2), q.ctypes.data_as(d2), rays[k:k + 1].ctypes.data)
    return rays


def simplify_scan(n, angle_increment):
    sel = (C.c_int * (n + 1))()
    inc = C.c_float(0)
    m = lib().og_simplify_scan(n, angle_increment, sel, n + 1, C.byref(inc))
    return list(sel[:m]), inc.value


def to_occupancy_grid(g, layer, data_min=0.0, data_max=255.0):
    out = np.empty(g.size[0] * g.size[1], np.int8)
    lib().og_to_occupancy_grid(C.byref(g), fptr(layer), data_min, data_max, out.ctypes.data)
    return out


def from_occupancy_grid(rows, cols, data):
    data = np.ascontiguousarray(data, np.int8)
    layer = np.empty(rows * cols, np.float32)
    lib().og_from_occupancy_grid(rows, cols, data.ctypes.data, fptr(layer))
    return layer


def hist_msg(hist, origin_hist, sector_angle):
    hist = np.ascontiguousarray(hist, np.float32)
    origin_hist = np.ascontiguousarray(origin_hist, np.float32)
    bins = len(hist) // 2
    x = np.empty(bins, np.uint16); y = np.empty(bins, np.uint16); yb = np.empty(bins, np.uint16); th = np.empty(2, np.uint16)
    n = lib().og_hist_msg(fptr(hist), fptr(origin_hist), len(hist), sector_angle, x.ctypes.data, y.ctypes.data,
                          yb.ctypes.data, th.ctypes.data)
    assert n == bins
    return x, y, yb, (int(th[0]), int(th[1]))


def tailor_plan(plan_xy, stride=5):
    plan = np.ascontiguousarray(plan_xy, np.float64).reshape(-1, 2)
    out = np.empty_like(plan)
    m = lib().og_tailor_plan(plan.ctypes.data_as(C.POINTER(C.c_double)), len(plan), stride,
                             out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:m].copy()


def follow_plan(plan_xy, plan_index, x, y, yaw):
    """(following, plan_index, desiredAngle, desiredDist) of Steerer::update (steerer.cpp:222-256)."""
    plan = np.ascontiguousarray(plan_xy, np.float64).reshape(-1, 2)
    idx = C.c_int(plan_index)
    out = (C.c_float * 2)()
    ok = lib().og_follow_plan(plan.ctypes.data_as(C.POINTER(C.c_double)), len(plan), C.byref(idx), x, y, yaw, out)
    return bool(ok), idx.value, np.float32(out[0]), np.float32(out[1])


def range_to_rays(readings):
    """RangeMapUpdater::bufferIncomingMsg (range_map_updater.cpp:38-76) for a batch of sonar readings."""
    rays = np.zeros(len(readings), RAY_DTYPE)
    for k, m in enumerate(readings):
        lib().og_range_to_ray(float(m["range"]), float(m["max_range"]), float(m["x"]), float(m["y"]), float(m["yaw"]),
                              rays[k:k + 1].ctypes.data_as(C.c_void_p))
    return rays
