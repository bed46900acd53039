"""Synthetic workloads of SURVEY.md section 8d (seeded numpy PCG64).  Pure input generation --
no algorithmic code lives here.  All layers are column-major float32 (linear = i + j*rows)."""
import numpy as np

from .capi import ASTAR_QUERY_DTYPE, POSE_DTYPE, RAY_DTYPE, RRT_QUERY_DTYPE


def occupancy_sparse(rows, cols, seed=1, occupied=0.02, unknown=0.10):
    """Config 2: 2 % of the cells hold a value from {30,60,...,180}, 10 % are NaN, the rest 0."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    m = np.zeros(n, np.float32)
    u = rng.random(n)
    occ = u < occupied
    m[occ] = rng.integers(1, 7, int(occ.sum())).astype(np.float32) * 30.0
    m[(u >= occupied) & (u < occupied + unknown)] = np.nan
    return m


def poses(n, length_x, length_y, seed=1, margin=1.5, speed=0, goal_dir=90.0, goal_dist=3000.0, dt=0.2):
    """Config 2 poses: x,y uniform in the inner (map - margin) square, yaw uniform in [-pi, pi)."""
    rng = np.random.default_rng(seed + 1000)
    p = np.zeros(n, POSE_DTYPE)
    p["x"] = rng.uniform(-(length_x / 2 - margin), length_x / 2 - margin, n)
    p["y"] = rng.uniform(-(length_y / 2 - margin), length_y / 2 - margin, n)
    p["yaw"] = rng.uniform(-np.pi, np.pi, n)
    p["dt"] = dt
    p["current_speed"] = speed
    p["goal_direction"] = goal_dir
    p["goal_distance"] = goal_dist
    p["goal_tolerance"] = 250.0
    return p


def obstacles_rect(rows, cols, density=0.30, seed=2, side=(4, 64), value=180.0):
    """Config 3: random rectangles (side 4..64 cells) up to `density`, plus a 1-cell border."""
    rng = np.random.default_rng(seed)
    m = np.zeros((cols, rows), np.float32)  # m[j, i]
    target = density * rows * cols
    filled = 0
    while filled < target:
        k = 256
        ws = rng.integers(side[0], side[1] + 1, k)
        hs = rng.integers(side[0], side[1] + 1, k)
        i0 = rng.integers(0, rows, k)
        j0 = rng.integers(0, cols, k)
        for a in range(k):
            blk = m[j0[a]:j0[a] + hs[a], i0[a]:i0[a] + ws[a]]
            filled += int(blk.size - np.count_nonzero(blk))
            blk[...] = value
            if filled >= target:
                break
    m[0, :] = value
    m[-1, :] = value
    m[:, 0] = value
    m[:, -1] = value
    return m.reshape(-1)


def free_component(master, rows, cols):
    """Boolean mask (column-major) of the largest 4-connected free region (free: NaN or <= 0)."""
    from scipy import ndimage
    free = ~(np.nan_to_num(master, nan=0.0) > 0.0)
    lab, n = ndimage.label(free.reshape(cols, rows))
    if n == 0:
        return np.zeros(rows * cols, bool)
    sizes = np.bincount(lab.reshape(-1))
    sizes[0] = 0
    return (lab == int(np.argmax(sizes))).reshape(-1)


def astar_queries(n, master, rows, cols, seed=2):
    """Config 3: (start, goal) uniform over the largest connected free region."""
    rng = np.random.default_rng(seed + 2000)
    cells = np.nonzero(free_component(master, rows, cols))[0]
    q = np.zeros(n, ASTAR_QUERY_DTYPE)
    q["start"] = rng.choice(cells, n)
    q["goal"] = rng.choice(cells, n)
    return q


def rays(n_poses, rays_per_pose, length_x, length_y, seed=4, lmin=1.0, lmax=6.0, hit=0.8, margin=6.5):
    """Config 5: `n_poses` robot origins, `rays_per_pose` rays each, bearings uniform, length uniform
    lmin..lmax metres, `hit` fraction end on an obstacle (ifClearEnd = false)."""
    rng = np.random.default_rng(seed)
    n = n_poses * rays_per_pose
    ox = np.repeat(rng.uniform(-(length_x / 2 - margin), length_x / 2 - margin, n_poses), rays_per_pose)
    oy = np.repeat(rng.uniform(-(length_y / 2 - margin), length_y / 2 - margin, n_poses), rays_per_pose)
    th = rng.uniform(-np.pi, np.pi, n)
    ln = rng.uniform(lmin, lmax, n)
    r = np.zeros(n, RAY_DTYPE)
    r["sx"], r["sy"] = ox, oy
    r["ex"], r["ey"] = ox + ln * np.cos(th), oy + ln * np.sin(th)
    r["clear_end"] = (rng.random(n) >= hit).astype(np.int32)
    return r


def rrt_queries(n, master, rows, cols, engine_get_position, seed=3, max_samples=200000):
    rng = np.random.default_rng(seed + 3000)
    cells = np.nonzero(free_component(master, rows, cols))[0]
    q = np.zeros(n, RRT_QUERY_DTYPE)
    s = rng.choice(cells, n)
    t = rng.choice(cells, n)
    for k in range(n):
        q["start"][k] = engine_get_position(int(s[k] % rows), int(s[k] // rows))
        q["target"][k] = engine_get_position(int(t[k] % rows), int(t[k] // rows))
    q["close_tolerance"] = 0.2
    q["seed"] = np.arange(1, n + 1, dtype=np.uint32)
    q["max_samples"] = max_samples
    return q


def laser_scans(n_scans, n_beams, length_x, length_y, seed=6, angle_increment=None, range_max=6.0, hit=0.8, moving=0.5):
    """Synthetic sensor_msgs/LaserScan batch (capi.SCAN_DTYPE descriptors + concatenated float32 ranges): sensor
    poses inside the map, 270-degree fans, ranges uniform in [range_min, range_max) for hits, range_max or +inf for
    misses, a few NaN / below-range_min returns (all dropped by the projection as in laser_geometry).  A share `moving`
    of the scans carries an end pose that differs from the start pose (the high-fidelity projection interpolates)."""
    from .capi import SCAN_DTYPE
    rng = np.random.default_rng(seed)
    scans = np.zeros(n_scans, SCAN_DTYPE)
    ranges = np.empty(n_scans * n_beams, np.float32)
    fan = np.float32(1.5 * np.pi)
    inc = np.float32(fan / n_beams) if angle_increment is None else np.float32(angle_increment)
    for k in range(n_scans):
        r = rng.uniform(0.12, range_max, n_beams).astype(np.float32)
        u = rng.random(n_beams)
        r[u > hit] = np.float32(range_max)
        r[u > hit + 0.1] = np.inf
        r[rng.random(n_beams) < 0.01] = np.nan
        r[rng.random(n_beams) < 0.01] = np.float32(0.01)
        ranges[k * n_beams:(k + 1) * n_beams] = r
        x, y, yaw = rng.uniform(-0.4, 0.4) * length_x, rng.uniform(-0.4, 0.4) * length_y, rng.uniform(-np.pi, np.pi)
        xe, ye, yawe = x, y, yaw
        if rng.random() < moving:     # the sensor moved during the scan: tf's end transform differs (yaw may cross +-pi)
            xe, ye = x + rng.uniform(-0.05, 0.05), y + rng.uniform(-0.05, 0.05)
            yawe = (yaw + rng.uniform(-0.2, 0.2) + np.pi) % (2 * np.pi) - np.pi
        scans[k] = (np.float32(-0.5 * fan), np.float32(-0.5 * fan + inc * n_beams), inc, np.float32(0.1), np.float32(range_max),
                    n_beams, k * n_beams, x, y, yaw, xe, ye, yawe)
    return scans, ranges


def laser_scans_tf(n_scans, n_beams, length_x, length_y, seed=7, angle_increment=None, range_max=6.0, tilt=0.35, planar=0.25, **kw):
    """laser_scans() with the sensor's FULL pose (capi.SCAN_TF_DTYPE: tf's translation + quaternion x y z w at both ends of the
    scan): mounts rolled / pitched by up to `tilt` rad and raised off the ground; a share `planar` of the scans keeps a pure
    yaw (the planar entry point's case).  Moving scans also change roll and pitch a little while they sweep."""
    from scipy.spatial.transform import Rotation
    from .capi import SCAN_TF_DTYPE
    base, ranges = laser_scans(n_scans, n_beams, length_x, length_y, seed=seed, angle_increment=angle_increment, range_max=range_max, **kw)
    rng = np.random.default_rng(seed + 1000)
    scans = np.zeros(n_scans, SCAN_TF_DTYPE)
    for f in ("angle_min", "angle_max", "angle_increment", "range_min", "range_max", "n_ranges", "ranges_offset"):
        scans[f] = base[f]
    for k in range(n_scans):
        flat = rng.random() < planar
        roll, pitch = (0.0, 0.0) if flat else rng.uniform(-tilt, tilt, 2)
        z = rng.uniform(0.1, 1.2)
        moving = base["yaw_end"][k] != base["yaw"][k] or base["x_end"][k] != base["x"][k]
        droll, dpitch, dz = (rng.uniform(-0.03, 0.03, 3) if (moving and not flat) else (0.0, 0.0, 0.0))
        q0 = Rotation.from_euler("ZYX", [base["yaw"][k], pitch, roll]).as_quat()
        q1 = Rotation.from_euler("ZYX", [base["yaw_end"][k], pitch + dpitch, roll + droll]).as_quat()
        if rng.random() < 0.5:
            q1 = -q1                 # the same rotation: tf's slerp has to take the short way round
        scans["t"][k] = (base["x"][k], base["y"][k], z)
        scans["t_end"][k] = (base["x_end"][k], base["y_end"][k], z + dz)
        scans["q"][k], scans["q_end"][k] = q0, q1
    return scans, ranges
