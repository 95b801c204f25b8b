"""Seeded synthetic inputs: there are no datasets on the build or GPU boxes.

Front-end streams: a large textured canvas (value-noise background, contrast rectangles, small
blobs) viewed through a window that moves along a Lissajous path — cheap to render, rich in FAST
corners at every pyramid level, deterministic from the seed.
"""
import numpy as np

EUROC = (752, 480)   # code/Examples/Monocular/EuRoC.yaml
KITTI = (1241, 376)  # code/Examples/Monocular/KITTI00-02.yaml


def _value_noise(rng, h, w, octaves=4, amplitude=40.0):
    out = np.zeros((h, w), np.float32)
    for o in range(octaves):
        cells = 4 << o
        g = rng.standard_normal((cells + 2, cells + 2)).astype(np.float32)
        ys = np.linspace(0, cells, h, endpoint=False)
        xs = np.linspace(0, cells, w, endpoint=False)
        y0 = ys.astype(int); x0 = xs.astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        out += (amplitude / (1 << o)) * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    return out


def make_canvas(seed, w, h, n_rect=None, n_blob=None):
    """A (h, w) uint8 textured canvas."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = 110.0 + _value_noise(rng, h, w)
    n_rect = n_rect if n_rect is not None else (w * h) // 2500
    n_blob = n_blob if n_blob is not None else (w * h) // 600
    for _ in range(n_rect):
        rw, rh = rng.integers(8, 90, 2)
        x, y = rng.integers(0, w - 8), rng.integers(0, h - 8)
        img[y:y + rh, x:x + rw] += rng.uniform(-70, 70)
    for _ in range(n_blob):
        s = rng.integers(2, 7)
        x, y = rng.integers(0, w - 8), rng.integers(0, h - 8)
        img[y:y + s, x:x + s] += rng.choice([-1.0, 1.0]) * rng.uniform(25, 90)
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_image(seed, size=EUROC):
    return make_canvas(seed, size[0], size[1])


class FrameStream:
    """Deterministic stream of (h, w) uint8 frames: a window sliding over a canvas.  With K and dist (k1 k2 p1 p2
    [k3]) the window is seen through that lens: pixel (u, v) of a frame shows the canvas where the pinhole camera K
    would have put the undistorted point of (u, v) (radial-tangential model, the same five fixed-point iterations as
    cv::undistortPoints; bilinear sampling), so undistorted keypoints are consistent with the planar scene."""

    def __init__(self, seed=20221001, size=EUROC, margin=160, K=None, dist=None):
        self.w, self.h = size
        self.margin = margin
        self.canvas = make_canvas(seed, self.w + 2 * margin, self.h + 2 * margin)
        self._map = None
        if K is not None and dist is not None and float(dist[0]) != 0.0:
            fx, fy, cx, cy = [float(v) for v in K]
            d = [float(v) for v in dist] + [0.0] * (5 - len(dist))
            k1, k2, p1, p2, k3 = d
            v, u = np.mgrid[0:self.h, 0:self.w].astype(np.float64)
            x0, y0 = (u - cx) / fx, (v - cy) / fy
            x, y = x0.copy(), y0.copy()
            for _ in range(5):
                r2 = x * x + y * y
                icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
                dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
                dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
                x, y = (x0 - dx) * icd, (y0 - dy) * icd
            self._map = ((fx * x + cx).astype(np.float32), (fy * y + cy).astype(np.float32))

    def offset(self, t):
        m = self.margin
        return (int(round(m + (m - 1) * np.sin(0.013 * t))), int(round(m + (m - 1) * np.sin(0.021 * t + 0.5))))

    def frame(self, t):
        ox, oy = self.offset(t)
        if self._map is None:
            return np.ascontiguousarray(self.canvas[oy:oy + self.h, ox:ox + self.w])
        H, W = self.canvas.shape
        sx = np.clip(self._map[0] + np.float32(ox), 0, W - 1.001)
        sy = np.clip(self._map[1] + np.float32(oy), 0, H - 1.001)
        x0, y0 = sx.astype(np.int32), sy.astype(np.int32)
        ax, ay = sx - x0, sy - y0
        c = self.canvas.astype(np.float32)
        top = c[y0, x0] * (1 - ax) + c[y0, x0 + 1] * ax
        bot = c[y0 + 1, x0] * (1 - ax) + c[y0 + 1, x0 + 1] * ax
        return np.clip(np.rint(top * (1 - ay) + bot * ay), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# matcher inputs: descriptors = base descriptors with Binomial(256, p) flipped bits,
# geometry = jittered keypoint positions.  Everything is returned as plain numpy arrays.
# ------------------------------------------------------------------------------------------------
SCALE_FACTORS = np.array([1.2 ** i for i in range(8)], np.float32)


def flip_bits(rng, desc, p):
    """Flip each of the 256 bits of every 32-byte descriptor with probability p."""
    mask = np.packbits(rng.random((desc.shape[0], 256)) < p, axis=1)
    return np.bitwise_xor(desc, mask)


def make_frame_arrays(rng, n, size=EUROC, levels=8, distort_margin=6.0, dup_frac=0.02):
    """Random undistorted keypoints (a few fall slightly outside the image like real undistorted points)."""
    w, h = size
    x = rng.uniform(-distort_margin, w + distort_margin, n).astype(np.float32)
    y = rng.uniform(-distort_margin, h + distort_margin, n).astype(np.float32)
    probs = (1 / 1.2) ** np.arange(levels)
    octave = rng.choice(levels, n, p=probs / probs.sum()).astype(np.int32)
    angle = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    ndup = int(n * dup_frac)  # duplicated descriptors -> equal distances -> exercises the tie-break order
    if ndup:
        src = rng.integers(0, n, ndup)
        dst = rng.integers(0, n, ndup)
        desc[dst] = desc[src]
    bounds = (0.0, float(w), 0.0, float(h))  # mnMinX, mnMaxX, mnMinY, mnMaxY (no distortion)
    return dict(x=x, y=y, octave=octave, angle=angle, desc=desc, bounds=bounds, scale_factors=SCALE_FACTORS)


def make_m1_case(seed, n_kp=1000, n_mp=2000, p_flip=0.15, size=EUROC, jitter=2.0, prebound_frac=0.1):
    """Frame + local map points for SearchByProjection(Frame&, vector<MapPoint*>&, th)."""
    rng = np.random.default_rng(seed)
    fr = make_frame_arrays(rng, n_kp, size)
    k = rng.integers(0, n_kp, n_mp)
    outlier = rng.random(n_mp) < 0.25
    proj_x = (fr["x"][k] + rng.normal(0, jitter, n_mp)).astype(np.float32)
    proj_y = (fr["y"][k] + rng.normal(0, jitter, n_mp)).astype(np.float32)
    proj_x[outlier] = rng.uniform(0, size[0], outlier.sum()).astype(np.float32)
    proj_y[outlier] = rng.uniform(0, size[1], outlier.sum()).astype(np.float32)
    desc = flip_bits(rng, fr["desc"][k], p_flip)
    desc[outlier] = rng.integers(0, 256, (int(outlier.sum()), 32)).astype(np.uint8)
    pred_level = np.clip(fr["octave"][k] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    view_cos = np.where(rng.random(n_mp) < 0.5, 0.9995, 0.9).astype(np.float32)
    mps = dict(in_view=(rng.random(n_mp) < 0.9).astype(np.uint8), proj_x=proj_x, proj_y=proj_y, view_cos=view_cos,
               pred_level=pred_level, desc=desc, has_obs=(rng.random(n_mp) < 0.97).astype(np.uint8))
    fr["excluded"] = (rng.random(n_kp) < prebound_frac).astype(np.uint8)
    return fr, mps


def make_m2_case(seed, n_kp=1000, n_last=1000, p_flip=0.1, size=EUROC, jitter=3.0):
    """Current frame + last frame's projected map points for SearchByProjection(cur, last, th, bMono)."""
    rng = np.random.default_rng(seed)
    fr = make_frame_arrays(rng, n_kp, size)
    k = rng.integers(0, n_kp, n_last)
    u = (fr["x"][k] + rng.normal(0, jitter, n_last)).astype(np.float32)
    v = (fr["y"][k] + rng.normal(0, jitter, n_last)).astype(np.float32)
    octave = np.clip(fr["octave"][k] + rng.integers(-1, 2, n_last), 0, 7).astype(np.int32)
    rot = rng.choice([3.0, 40.0, 200.0], n_last, p=[0.8, 0.15, 0.05])  # a dominant rotation + outliers
    angle = ((fr["angle"][k] + rot + rng.normal(0, 2, n_last)) % 360).astype(np.float32)
    last = dict(valid=(rng.random(n_last) < 0.6).astype(np.uint8), u=u, v=v, octave=octave, angle=angle,
                desc=flip_bits(rng, fr["desc"][k], p_flip), has_obs=(rng.random(n_last) < 0.97).astype(np.uint8))
    fr["excluded"] = (rng.random(n_kp) < 0.05).astype(np.uint8)
    return fr, last


def make_m4_case(seed, n_kp=2000, p_flip=0.05, size=EUROC, shift=(12.0, -7.0)):
    """Two frames for SearchForInitialization (F2 = F1 moved by `shift` + noise, partly re-ordered)."""
    rng = np.random.default_rng(seed)
    f1 = make_frame_arrays(rng, n_kp, size, dup_frac=0.01)
    perm = rng.permutation(n_kp)
    f2 = dict(x=(f1["x"][perm] + shift[0] + rng.normal(0, 0.7, n_kp)).astype(np.float32),
              y=(f1["y"][perm] + shift[1] + rng.normal(0, 0.7, n_kp)).astype(np.float32),
              octave=f1["octave"][perm].copy(),
              angle=((f1["angle"][perm] + 5 + rng.normal(0, 3, n_kp)) % 360).astype(np.float32),
              desc=flip_bits(rng, f1["desc"][perm], p_flip), bounds=f1["bounds"],
              scale_factors=f1["scale_factors"])
    prev = np.stack([f1["x"], f1["y"]], 1).astype(np.float32)
    return f1, f2, prev


# ------------------------------------------------------------------------------------------------
# bundle-adjustment windows: LBA-S / LBA-M / LBA-L / GBA-1 / GBA-2
# ------------------------------------------------------------------------------------------------
EUROC_K = (458.654, 457.296, 367.215, 248.375)  # code/Examples/Monocular/EuRoC.yaml
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)  # code/Examples/Monocular/KITTI00-02.yaml
EUROC_DIST = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)  # k1 k2 p1 p2, code/Examples/Monocular/EuRoC.yaml
BA_CASES = {  # name: (free KFs, fixed KFs, points)
    "LBA-S": (8, 10, 800), "LBA-M": (25, 40, 3000), "LBA-L": (40, 60, 6000),
    "GBA-1": (299, 1, 30000), "GBA-2": (1499, 1, 120000),
}


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_ba_problem(seed, n_free=8, n_fixed=10, n_points=800, size=EUROC, K=EUROC_K, pixel_sigma=1.0,
                    outlier_frac=0.05, pose_noise=(0.02, 0.5), point_noise=0.03, max_obs=None, max_yaw=None):
    """A seeded local-BA window: cameras on an arc looking at a point cloud; returns the flattened problem
    (float32 like the map stores it) plus the generating ground truth.
    The cameras turn by 0.02 rad per keyframe: beyond ~150 keyframes the ones at the ends of the arc look away from the
    cloud and observe nothing - g2o gives such keyframes no Hessian index, and neither does so_bundle_adjust
    (so_ba_info.n_free_keyframes counts the ones that have one).  max_yaw (rad) caps the total turn so that EVERY
    keyframe sees points - what the window-size sweeps want (tools/lba_bench.py --sweep)."""
    rng = np.random.default_rng(seed)
    n_poses = n_free + n_fixed
    w, h = size
    fx, fy, cx, cy = K
    Rs, ts = [], []
    for i in range(n_poses):  # camera centres on a slowly advancing arc, all looking towards +z
        c = np.array([0.25 * i - 0.125 * n_poses + rng.normal(0, 0.05), rng.normal(0, 0.08), rng.normal(0, 0.1)])
        yaw_rate = 0.02 if max_yaw is None else min(0.02, 2.0 * float(max_yaw) / max(n_poses, 1))
        R = _rodrigues(np.array([rng.normal(0, 0.03), yaw_rate * (i - n_poses / 2) + rng.normal(0, 0.02),
                                 rng.normal(0, 0.02)]))
        Rs.append(R)  # Rcw
        ts.append(-R @ c)
    X = np.stack([rng.uniform(-0.15 * n_poses - 2, 0.15 * n_poses + 2, n_points), rng.uniform(-2, 2, n_points),
                  rng.uniform(3, 12, n_points)], 1)
    scale = 1.2 ** np.arange(8)
    e_pose, e_pt, obs, inv_s2, is_out = [], [], [], [], []
    order = np.arange(n_poses)
    for j in range(n_points):
        rng.shuffle(order)  # the reference iterates a std::map<KeyFrame*,...>: arbitrary keyframe order
        seen = 0
        cap = int(rng.integers(3, 11)) if max_obs == "auto" else max_obs
        for i in order:
            pc = Rs[i] @ X[j] + ts[i]
            if pc[2] < 0.5:
                continue
            u, v = fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy
            if not (20 < u < w - 20 and 20 < v < h - 20):
                continue
            if cap is not None and seen >= cap:
                break
            octave = int(np.clip(np.floor(np.log(pc[2] / 3.0) / np.log(1.2)), 0, 7))
            noise = rng.normal(0, pixel_sigma * scale[octave], 2)
            out = rng.random() < outlier_frac
            if out:
                noise = rng.uniform(-40, 40, 2)
            e_pose.append(i); e_pt.append(j); obs.append([u + noise[0], v + noise[1]])
            inv_s2.append(1.0 / (scale[octave] ** 2)); is_out.append(out)
            seen += 1
    e_pose, e_pt = np.array(e_pose, np.int32), np.array(e_pt, np.int32)
    # poses sorted by "mnId": free keyframes are the most recent ones -> highest ids; keep the first one fixed
    fixed = np.zeros(n_poses, np.uint8)
    fixed[:n_fixed] = 1
    Tcw_gt = np.stack([np.hstack([Rs[i], ts[i][:, None]]).reshape(12) for i in range(n_poses)])
    Tcw0 = Tcw_gt.copy()
    for i in range(n_poses):
        if fixed[i]:
            continue
        dR = _rodrigues(rng.normal(0, np.deg2rad(pose_noise[1]), 3))
        T = np.hstack([dR @ Rs[i], (dR @ ts[i] + rng.normal(0, pose_noise[0], 3))[:, None]])
        Tcw0[i] = T.reshape(12)
    X0 = X + rng.normal(0, point_noise, X.shape)
    return dict(Tcw=Tcw0.astype(np.float32), fixed=fixed, intr=np.tile(np.array(K, np.float32), (n_poses, 1)),
                Xw=X0.astype(np.float32), edge_pose=e_pose, edge_point=e_pt, obs=np.array(obs, np.float32),
                inv_sigma2=np.array(inv_s2, np.float32), gt_Tcw=Tcw_gt, gt_Xw=X, gt_outlier=np.array(is_out, bool))


def make_multiagent_map(seed, n_agents=8, kfs_per_agent=187, n_points=120000, spacing=0.4, view_range=10.0, size=EUROC,
                        K=EUROC_K, pixel_sigma=1.0, outlier_frac=0.05, pose_noise=(0.02, 0.5), point_noise=0.03):
    """A merged multi-agent map for global bundle adjustment (BASELINE configs[4]) with the structure real SLAM maps
    have: every agent drives its own street of a grid (half of them along x, half along y, forward-looking camera,
    one keyframe every `spacing` metres), map points line the streets, a keyframe sees a point only inside the image
    and within `view_range` metres, and a point keeps 3-10 of the keyframes that see it.  Covisibility is therefore
    banded along each trajectory and the agents are linked only where their streets cross - unlike make_ba_problem,
    whose cameras all look at one cloud (reduced camera system nearly dense).  Keyframes are numbered agent by agent
    (the order a merged map has: mnId + 1e6 * mapId), only the very first one is fixed.  Same return layout as
    make_ba_problem.  Vectorised: a 1500-keyframe / 120 k-point map takes a few seconds."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    w, h = size
    fx, fy, cx, cy = K
    na_x = (n_agents + 1) // 2      # agents 0 .. na_x-1 drive along x, the others along y
    spacing = max(spacing, 45.0 / kfs_per_agent)
    length = spacing * kfs_per_agent
    D = max(0.53 * length, 2.4 * view_range)  # distance between parallel x-streets: out of each other's sight
    dx = 1.5 * view_range                     # distance between parallel y-streets
    centres, Rs, street = [], [], []
    for a in range(n_agents):
        t = spacing * np.arange(kfs_per_agent) + rng.normal(0, 0.03, kfs_per_agent)
        lat = np.cumsum(rng.normal(0, 0.01, kfs_per_agent))  # slow lateral drift
        hgt = rng.normal(0, 0.03, kfs_per_agent)
        if a < na_x:     # x-street i at y = D i, x in [0, length]
            org, fwd = np.array([0.0, D * a, 0.0]), np.array([1.0, 0.0, 0.0])
        else:            # y-street j crosses the x-streets (j mod (na_x - 1)) and the next one
            j = a - na_x
            i0 = j % max(na_x - 1, 1)
            org = np.array([0.2 * length + dx * j, D * i0 - 0.23 * length, 0.0])
            fwd = np.array([0.0, 1.0, 0.0])
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        c = org[None, :] + t[:, None] * fwd[None, :] + lat[:, None] * right[None, :] + hgt[:, None] * up[None, :]
        base = np.stack([right, -up, fwd])  # camera axes (x right, y down, z forward) as rows: Rcw
        for i in range(kfs_per_agent):
            Rs.append(_rodrigues(rng.normal(0, 0.02, 3)) @ base)
        centres.append(c)
        street.append((org, fwd, right))
    centres = np.concatenate(centres)
    Rs = np.stack(Rs)
    n_poses = len(centres)
    ts = -np.einsum("nij,nj->ni", Rs, centres)
    # points along the streets: lateral offset 1.5-4 m on either side, height -1.5..3 m
    which = rng.integers(0, n_agents, n_points)
    s_along = rng.uniform(-2.0, length + view_range * 0.5, n_points)
    lateral = rng.uniform(1.5, 4.0, n_points) * rng.choice([-1.0, 1.0], n_points)
    height = rng.uniform(-1.5, 3.0, n_points)
    orgs = np.stack([st[0] for st in street]); fwds = np.stack([st[1] for st in street]); rights = np.stack([st[2] for st in street])
    X = orgs[which] + s_along[:, None] * fwds[which] + lateral[:, None] * rights[which]
    X[:, 2] += height
    tree = cKDTree(X)
    scale = 1.2 ** np.arange(8)
    ev_pose, ev_pt, ev_uv, ev_z = [], [], [], []
    for i in range(n_poses):
        idx = np.asarray(tree.query_ball_point(centres[i], view_range), np.int64)
        if len(idx) == 0:
            continue
        pc = X[idx] @ Rs[i].T + ts[i]
        ok = pc[:, 2] > 0.5
        u = fx * pc[:, 0] / np.where(ok, pc[:, 2], 1.0) + cx
        v = fy * pc[:, 1] / np.where(ok, pc[:, 2], 1.0) + cy
        ok &= (u > 20) & (u < w - 20) & (v > 20) & (v < h - 20)
        sel = np.nonzero(ok)[0]
        ev_pose.append(np.full(len(sel), i, np.int32)); ev_pt.append(idx[sel].astype(np.int32))
        ev_uv.append(np.stack([u[sel], v[sel]], 1)); ev_z.append(pc[sel, 2])
    ev_pose = np.concatenate(ev_pose); ev_pt = np.concatenate(ev_pt)
    ev_uv = np.concatenate(ev_uv); ev_z = np.concatenate(ev_z)
    # every point keeps a random 3-10 of its observations (at least 2, else it is dropped)
    key = rng.random(len(ev_pt))
    order = np.lexsort((key, ev_pt))
    ev_pose, ev_pt, ev_uv, ev_z = ev_pose[order], ev_pt[order], ev_uv[order], ev_z[order]
    start = np.searchsorted(ev_pt, np.arange(n_points))
    count = np.diff(np.append(start, len(ev_pt)))
    rank = np.arange(len(ev_pt)) - start[ev_pt]
    cap = rng.integers(3, 11, n_points)
    keep = (rank < cap[ev_pt]) & (count[ev_pt] >= 2)
    ev_pose, ev_pt, ev_uv, ev_z = ev_pose[keep], ev_pt[keep], ev_uv[keep], ev_z[keep]
    used = np.unique(ev_pt)
    remap = np.full(n_points, -1, np.int64)
    remap[used] = np.arange(len(used))
    ev_pt = remap[ev_pt].astype(np.int32)
    X = X[used]
    # the reference iterates a std::map<KeyFrame*, ...>: observation order inside a point is arbitrary
    shuffle = np.lexsort((rng.random(len(ev_pt)), ev_pt))
    ev_pose, ev_pt, ev_uv, ev_z = ev_pose[shuffle], ev_pt[shuffle], ev_uv[shuffle], ev_z[shuffle]
    octave = np.clip(np.floor(np.log(ev_z / 3.0) / np.log(1.2)), 0, 7).astype(int)
    noise = rng.normal(0, pixel_sigma, (len(ev_pt), 2)) * scale[octave][:, None]
    is_out = rng.random(len(ev_pt)) < outlier_frac
    noise[is_out] = rng.uniform(-40, 40, (int(is_out.sum()), 2))
    fixed = np.zeros(n_poses, np.uint8)
    fixed[0] = 1
    Tcw_gt = np.concatenate([Rs, ts[:, :, None]], 2).reshape(n_poses, 12)
    Tcw0 = Tcw_gt.copy()
    for i in range(1, n_poses):
        dR = _rodrigues(rng.normal(0, np.deg2rad(pose_noise[1]), 3))
        Tcw0[i] = np.hstack([dR @ Rs[i], (dR @ ts[i] + rng.normal(0, pose_noise[0], 3))[:, None]]).reshape(12)
    X0 = X + rng.normal(0, point_noise, X.shape)
    return dict(Tcw=Tcw0.astype(np.float32), fixed=fixed, intr=np.tile(np.array(K, np.float32), (n_poses, 1)),
                Xw=X0.astype(np.float32), edge_pose=ev_pose.astype(np.int32), edge_point=ev_pt,
                obs=(ev_uv + noise).astype(np.float32), inv_sigma2=(1.0 / scale[octave] ** 2).astype(np.float32),
                gt_Tcw=Tcw_gt, gt_Xw=X, gt_outlier=is_out, agent_of_pose=np.repeat(np.arange(n_agents), kfs_per_agent))


MULTIAGENT_CASES = {  # name: (agents, keyframes per agent, candidate points)
    "GBA-1r": (4, 75, 30000), "GBA-2r": (8, 188, 120000), "GBA-4k": (8, 512, 330000),
}


def make_ba_case(name, seed=0, **kw):
    if name in MULTIAGENT_CASES:
        na, nk, npnt = MULTIAGENT_CASES[name]
        return make_multiagent_map(seed, na, nk, npnt, **kw)

    nf, nx, npnt = BA_CASES[name]
    kw.setdefault("max_obs", "auto")  # ~6.5 observations per point -> the edge counts of SURVEY.md 8d
    return make_ba_problem(seed, nf, nx, npnt, **kw)


def make_bow_case(seed, n1=1000, n2=1000, n_nodes=100, p_flip=0.1):
    """Two keyframes with vocabulary-node labels for SearchByBoW / SearchForTriangulation."""
    rng = np.random.default_rng(seed)
    f2 = make_frame_arrays(rng, n2, dup_frac=0.02)
    src = rng.integers(0, n2, n1)
    desc1 = flip_bits(rng, f2["desc"][src], p_flip)
    node2 = rng.integers(0, n_nodes, n2) * 7 + 3  # arbitrary increasing ids with gaps
    node1 = node2[src].copy()
    stray = rng.random(n1) < 0.1  # quantised into a different node
    node1[stray] = rng.integers(0, n_nodes + 20, int(stray.sum())) * 7 + 3
    kf1 = dict(x=f2["x"][src] + rng.normal(0, 2, n1).astype(np.float32), y=f2["y"][src] + rng.normal(0, 2, n1).astype(np.float32),
               angle=((f2["angle"][src] + rng.choice([4.0, 90.0], n1, p=[0.85, 0.15])) % 360).astype(np.float32),
               desc=desc1, valid=(rng.random(n1) < 0.7).astype(np.uint8), free=(rng.random(n1) < 0.6).astype(np.uint8))
    kf2 = dict(x=f2["x"], y=f2["y"], octave=f2["octave"], angle=f2["angle"], desc=f2["desc"],
               valid=(rng.random(n2) < 0.8).astype(np.uint8), free=(rng.random(n2) < 0.7).astype(np.uint8))
    return kf1, node1, kf2, node2, src


def make_window_queries(seed, fr, nq, jitter=2.0, p_flip=0.12, th=3.0):
    """Projected map points for the Fuse / SearchBySim3 / SearchByProjection(KF, ...) window searches."""
    rng = np.random.default_rng(seed)
    n = len(fr["x"])
    k = rng.integers(0, n, nq)
    pred = np.clip(fr["octave"][k] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    return dict(valid=(rng.random(nq) < 0.85).astype(np.uint8),
                u=(fr["x"][k] + rng.normal(0, jitter, nq)).astype(np.float32),
                v=(fr["y"][k] + rng.normal(0, jitter, nq)).astype(np.float32),
                radius=(th * SCALE_FACTORS[pred]).astype(np.float32), pred_level=pred,
                min_level=(pred - 1).astype(np.int32), max_level=(pred + rng.integers(0, 2, nq)).astype(np.int32),
                desc=flip_bits(rng, fr["desc"][k], p_flip), angle=((fr["angle"][k] + 3) % 360).astype(np.float32))


def make_pose_case(seed, n=300, K=EUROC_K, size=EUROC, pixel_sigma=1.0, outlier_frac=0.1, pose_noise=(0.05, 1.5)):
    """Frame pose + matched map points for Optimizer::PoseOptimization."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    R = _rodrigues(rng.normal(0, 0.1, 3))
    t = rng.normal(0, 0.3, 3)
    uv = np.stack([rng.uniform(30, size[0] - 30, n), rng.uniform(30, size[1] - 30, n)], 1)
    z = rng.uniform(2, 12, n)
    pc = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)
    Xw = (R.T @ (pc - t).T).T
    octave = np.clip(np.floor(np.log(z / 2.0) / np.log(1.2)), 0, 7).astype(int)
    scale = 1.2 ** octave
    obs = uv + rng.normal(0, pixel_sigma, (n, 2)) * scale[:, None]
    out = rng.random(n) < outlier_frac
    obs[out] += rng.uniform(-50, 50, (int(out.sum()), 2))
    dR = _rodrigues(rng.normal(0, np.deg2rad(pose_noise[1]), 3))
    T0 = np.hstack([dR @ R, (dR @ t + rng.normal(0, pose_noise[0], 3))[:, None]]).reshape(12)
    return dict(Tcw=T0.astype(np.float32), intr=np.array(K, np.float32), Xw=Xw.astype(np.float32),
                obs=obs.astype(np.float32), inv_sigma2=(1.0 / scale ** 2).astype(np.float32),
                gt_Tcw=np.hstack([R, t[:, None]]).reshape(12), gt_outlier=out)


def make_frustum_case(seed, n=4000, K=EUROC_K, size=EUROC):
    """A frame pose and local map points for Frame::isInFrustum: points in front / behind / outside the image,
    inside / outside their scale-invariance range, seen frontally / at grazing angles."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    R = _rodrigues(rng.normal(0, 0.3, 3))
    t = rng.normal(0, 1.0, 3)
    Tcw = np.hstack([R, t[:, None]]).astype(np.float32)
    uv = np.stack([rng.uniform(-150, size[0] + 150, n), rng.uniform(-150, size[1] + 150, n)], 1)
    z = rng.uniform(0.5, 20, n) * np.where(rng.random(n) < 0.08, -1, 1)
    pc = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)
    Xw = (R.T @ (pc - t).T).T.astype(np.float32)
    Ow = -R.T @ t
    view = Xw - Ow
    view /= np.linalg.norm(view, axis=1, keepdims=True) + 1e-12
    tilt = np.stack([_rodrigues(rng.normal(0, 0.6, 3)) @ v for v in view])
    normal = tilt.astype(np.float32)
    d = np.linalg.norm(Xw - Ow, axis=1)
    max_dist = (d * rng.uniform(0.6, 3.0, n)).astype(np.float32)
    min_dist = (max_dist / (1.2 ** 7) * rng.uniform(0.8, 1.3, n)).astype(np.float32)
    return dict(Tcw=Tcw.reshape(12), Xw=Xw, normal=normal, max_dist=max_dist, min_dist=min_dist)


def make_kf_store_case(seed, n_agents=3, kfs_per_agent=6, n_kp=300, bound_frac=0.5, n_places=None, p_flip=0.04,
                       revisit_frac=0.6, ragged=True):
    """Keyframes of several agents for the cross-agent candidate search: every keyframe looks at one of
    `n_places` places; a place is a pool of descriptors with orientations, a keyframe sees a random subset of it
    under bit noise (two views of one feature differ by ~2 p 256 bits) plus clutter, in random keypoint order, rotated
    as a whole.  Returns a list of dicts (agent, keyframe_id, desc, angle, xy, octave, map_point_id, valid, place) in the
    order an exchange would deliver them (time-major: keyframe t of every agent, then t + 1)."""
    rng = np.random.default_rng(seed)
    n_places = n_places or max(2, (n_agents * kfs_per_agent) // 3)
    pool = [dict(desc=rng.integers(0, 256, (n_kp, 32)).astype(np.uint8), angle=rng.uniform(0, 360, n_kp).astype(np.float32))
            for _ in range(n_places)]
    out = []
    for t in range(kfs_per_agent):
        for a in range(n_agents):
            place = int(rng.integers(0, n_places))
            n = int(rng.integers(max(1, n_kp // 2), n_kp + 1)) if ragged else n_kp
            n_seen = int(revisit_frac * n)
            pick = rng.permutation(n_kp)[:n_seen]
            desc = np.concatenate([flip_bits(rng, pool[place]["desc"][pick], p_flip),
                                   rng.integers(0, 256, (n - n_seen, 32)).astype(np.uint8)])
            roll = np.float32(rng.uniform(0, 360))
            angle = np.concatenate([pool[place]["angle"][pick] + roll + rng.normal(0, 2.0, n_seen).astype(np.float32),
                                    rng.uniform(0, 360, n - n_seen).astype(np.float32)])
            angle = np.mod(angle, np.float32(360)).astype(np.float32)
            order = rng.permutation(n)
            desc, angle = np.ascontiguousarray(desc[order]), np.ascontiguousarray(angle[order])
            seen = np.zeros(n, bool)
            seen[:n_seen] = True
            seen = seen[order]
            # bindings: most of the re-observed features carry a map point, some clutter does too
            bound = (seen & (rng.random(n) < min(1.0, bound_frac * 1.5))) | (~seen & (rng.random(n) < bound_frac * 0.3))
            mp = np.where(bound, rng.integers(0, 1 << 30, n), -1).astype(np.int32)
            out.append(dict(agent=a, keyframe_id=1000 * a + t, desc=desc, angle=angle,
                            xy=rng.uniform(0, 752, (n, 2)).astype(np.float32), octave=rng.integers(0, 8, n).astype(np.int32),
                            map_point_id=mp, valid=(mp >= 0).astype(np.uint8), place=place,
                            Tcw=rng.normal(size=12).astype(np.float32)))
    return out


# ---------------------------------------------------------------------------------------------------------------
# Keyframe-side map-point searches: a target keyframe, a pose (or Sim3) and map points whose
# projections land near its keypoints - with every gate of the reference's loops exercised: points behind the camera,
# outside the image, outside their scale-invariance range, seen at grazing angles, flagged bad / already found.
# ---------------------------------------------------------------------------------------------------------------
LOG_SCALE_FACTOR = float(np.log(np.float32(1.2)))  # mfLogScaleFactor = log(mfScaleFactor), code/src/Frame.cc:201
INV_LEVEL_SIGMA2 = (1.0 / (SCALE_FACTORS.astype(np.float32) ** 2)).astype(np.float32)  # mvInvLevelSigma2


def _unproject(K, uv, z):
    fx, fy, cx, cy = K
    return np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)


def _map_points_for(rng, fr, k, Pc, R, t, octave_of, p_flip, level_spread=(-0.6, 0.5), bad_frac=0.1):
    """Map-point fields for camera-frame points Pc seen near keypoints k: world position through (R, t), a normal around
    the viewing ray (some grazing), a scale-invariance range that predicts a level around the keypoint's octave (some
    out of range), the keypoint's descriptor with flipped bits."""
    n = len(Pc)
    Xw = (R.T @ (Pc - t).T).T
    Ow = -R.T @ t
    view = Xw - Ow
    d = np.linalg.norm(view, axis=1)
    view /= d[:, None] + 1e-12
    tilt = np.where(rng.random(n) < 0.15, 1.2, 0.35)
    normal = np.stack([_rodrigues(rng.normal(0, s, 3)) @ v for v, s in zip(view, tilt)])
    max_dist = d * 1.2 ** (octave_of + rng.uniform(level_spread[0], level_spread[1], n))
    max_dist *= np.where(rng.random(n) < 0.06, 0.3, 1.0)   # beyond 1.2 * mfMaxDistance
    min_dist = max_dist / (1.2 ** 7) * rng.uniform(0.8, 1.25, n)
    min_dist *= np.where(rng.random(n) < 0.04, 400.0, 1.0)  # closer than 0.8 * mfMinDistance
    return dict(Xw=Xw.astype(np.float32), normal=normal.astype(np.float32), max_dist=max_dist.astype(np.float32),
                min_dist=min_dist.astype(np.float32), desc=flip_bits(rng, fr["desc"][k], p_flip),
                valid=(rng.random(n) >= bad_frac).astype(np.uint8))


EUROC_FRAME_BOUNDS = (-27.3418, 779.6142, -18.7226, 498.4311)  # like Frame::ComputeImageBounds under EuRoC's distortion


def as_keyframe_bounds(fr, frame_bounds=EUROC_FRAME_BOUNDS):
    """Give the frame arrays a KeyFrame's bounds: `bounds` = the Frame's float bounds truncated to int (code/include/
    KeyFrame.h:220), `grid_bounds` = the float ones the copied grid was filled with (code/src/KeyFrame.cc:58-72)."""
    fr["grid_bounds"] = tuple(float(np.float32(b)) for b in frame_bounds)
    fr["bounds"] = tuple(float(int(np.float32(b))) for b in frame_bounds)
    return fr


def make_projection_case(seed, n_kp=1000, n_mp=1200, size=EUROC, K=EUROC_K, p_flip=0.1, jitter=2.5, sim3_scale=None,
                         prebound_frac=0.0, keyframe_bounds=False):
    """Target keyframe + pose + map points for Fuse / SearchByProjection(KF, Scw) / SearchByProjection(Frame, KF).
    sim3_scale: the returned "Scw" is [s R | s t] (the pose itself is [R | t]); prebound_frac: keypoints already bound
    on entry (vpMatched[k] / mvpMapPoints[k] non-null)."""
    rng = np.random.default_rng(seed)
    fr = make_frame_arrays(rng, n_kp, size=size, distort_margin=25.0 if keyframe_bounds else 6.0)
    if keyframe_bounds:
        as_keyframe_bounds(fr)
    if prebound_frac > 0:
        fr["excluded"] = (rng.random(n_kp) < prebound_frac).astype(np.uint8)
    R = _rodrigues(rng.normal(0, 0.4, 3))
    t = rng.normal(0, 1.5, 3)
    k = rng.integers(0, n_kp, n_mp)
    uv = np.stack([fr["x"][k], fr["y"][k]], 1) + rng.normal(0, jitter, (n_mp, 2))
    far = rng.random(n_mp) < 0.08  # projections outside the image
    uv[far] = np.stack([rng.uniform(-200, size[0] + 200, far.sum()), rng.uniform(-200, size[1] + 200, far.sum())], 1)
    z = rng.uniform(0.8, 15.0, n_mp) * np.where(rng.random(n_mp) < 0.06, -1.0, 1.0)
    Pc = _unproject(K, uv, z)
    mp = _map_points_for(rng, fr, k, Pc, R, t, fr["octave"][k].astype(np.float64), p_flip)
    mp["angle"] = ((fr["angle"][k] + rng.normal(0, 4.0, n_mp)) % 360).astype(np.float32)  # pKF->mvKeysUn[i].angle
    s = 1.0 if sim3_scale is None else float(sim3_scale)
    T = np.hstack([R, t[:, None]])
    S = np.hstack([s * R, s * t[:, None]])
    return dict(frame=fr, Tcw=T.astype(np.float32).reshape(12), Scw=S.astype(np.float32).reshape(12), mp=mp,
                cam=K, log_scale_factor=LOG_SCALE_FACTOR, inv_level_sigma2=INV_LEVEL_SIGMA2)


def make_sim3_pair_case(seed, n=900, size=EUROC, K=EUROC_K, p_flip=0.08, s12=1.3, already_frac=0.1, keyframe_bounds=False):
    """Two keyframes of two maps related by a similarity (SearchBySim3, code/src/ORBmatcher.cc:1011-1221): points seen
    by keyframe 1 at its keypoints are seen by keyframe 2 where the Sim3 puts them (keypoints of 2 are shuffled), each
    keyframe's map points live in its own world frame."""
    rng = np.random.default_rng(seed)
    w, h = size
    fx, fy, cx, cy = K
    fr1 = make_frame_arrays(rng, n, size=size, dup_frac=0.0)
    z1 = rng.uniform(2.0, 9.0, n)
    Pc1 = _unproject(K, np.stack([fr1["x"], fr1["y"]], 1).astype(np.float64), z1)
    R12 = _rodrigues(rng.normal(0, 0.08, 3))
    t12 = rng.normal(0, 0.25, 3)
    Pc2 = (R12.T @ (Pc1 - t12).T).T / s12  # p_c2 = sR21 p_c1 + t21
    u2 = np.stack([fx * Pc2[:, 0] / Pc2[:, 2] + cx, fy * Pc2[:, 1] / Pc2[:, 2] + cy], 1) + rng.normal(0, 1.2, (n, 2))
    out = (Pc2[:, 2] <= 0) | (u2[:, 0] < 0) | (u2[:, 0] >= w) | (u2[:, 1] < 0) | (u2[:, 1] >= h)
    u2[out] = np.stack([rng.uniform(0, w, out.sum()), rng.uniform(0, h, out.sum())], 1)
    d2 = np.linalg.norm(Pc2, axis=1)
    d1 = np.linalg.norm(Pc1, axis=1)
    # octave of the point in keyframe 2 follows the distance ratio (closer by 1 / s12 -> coarser level)
    oct2 = np.clip(np.round(fr1["octave"] + np.log(d1 / d2) / np.log(1.2)), 0, 7).astype(np.int32)
    perm = rng.permutation(n)  # keypoint j of keyframe 2 shows point perm[j]
    fr2 = dict(x=u2[perm, 0].astype(np.float32), y=u2[perm, 1].astype(np.float32), octave=oct2[perm],
               angle=((fr1["angle"][perm] + 5) % 360).astype(np.float32), desc=flip_bits(rng, fr1["desc"][perm], p_flip),
               bounds=fr1["bounds"], scale_factors=SCALE_FACTORS)
    if keyframe_bounds:
        as_keyframe_bounds(fr1)
        as_keyframe_bounds(fr2)
    R1 = _rodrigues(rng.normal(0, 0.5, 3)); t1 = rng.normal(0, 2.0, 3)
    R2 = _rodrigues(rng.normal(0, 0.5, 3)); t2 = rng.normal(0, 2.0, 3)
    lv = (-0.5, 0.4)
    # map points of keyframe 1 are searched in keyframe 2: their range must predict the octave seen there, and v.v.
    mp1 = _map_points_for(rng, fr1, np.arange(n), Pc1, R1, t1, oct2 + np.log(d2 / d1) / np.log(1.2), p_flip, lv, already_frac)
    # (PredictScale uses mfMaxDistance / |p3Dc2|: max_dist above is d1-based, so shift it by the distance ratio)
    mp2 = _map_points_for(rng, fr2, np.arange(n), Pc2[perm], R2, t2,
                          fr1["octave"][perm] + np.log(d1[perm] / d2[perm]) / np.log(1.2), p_flip, lv, already_frac)
    return dict(frame1=fr1, frame2=fr2, T1w=np.hstack([R1, t1[:, None]]).astype(np.float32).reshape(12),
                T2w=np.hstack([R2, t2[:, None]]).astype(np.float32).reshape(12), s12=np.float32(s12),
                R12=R12.astype(np.float32).reshape(9), t12=t12.astype(np.float32), mp1=mp1, mp2=mp2, perm=perm, cam=K,
                log_scale_factor=LOG_SCALE_FACTOR)


def make_triangulation_case(seed, n=1500, K=EUROC_K, size=EUROC, baseline=0.6, pixel_sigma=0.7):
    """Two keyframes and their matched keypoints for LocalMapping::CreateNewMapPoints' per-match body: real 3-D points
    seen by both (pixel noise), plus wrong matches (large reprojection error), low-parallax pairs (far points), points
    behind a camera and octave pairs that break the scale consistency."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    R1 = _rodrigues(rng.normal(0, 0.2, 3)); t1 = rng.normal(0, 1.0, 3)
    dR = _rodrigues(rng.normal(0, 0.05, 3))
    R2 = dR @ R1
    t2 = dR @ t1 + np.array([baseline, 0.05 * baseline, 0.1 * baseline]) * rng.choice([-1, 1])
    uv1 = np.stack([rng.uniform(20, size[0] - 20, n), rng.uniform(20, size[1] - 20, n)], 1)
    z = rng.uniform(2.0, 12.0, n)
    far = rng.random(n) < 0.12
    z[far] = rng.uniform(300.0, 3000.0, far.sum())          # parallax below the 0.9998 bound
    Pc1 = _unproject(K, uv1, z)
    Xw = (R1.T @ (Pc1 - t1).T).T
    Pc2 = (R2 @ Xw.T).T + t2
    uv2 = np.stack([fx * Pc2[:, 0] / Pc2[:, 2] + cx, fy * Pc2[:, 1] / Pc2[:, 2] + cy], 1)
    uv1n = uv1 + rng.normal(0, pixel_sigma, (n, 2))
    uv2n = uv2 + rng.normal(0, pixel_sigma, (n, 2))
    wrong = rng.random(n) < 0.1
    uv2n[wrong] += rng.normal(0, 25.0, (wrong.sum(), 2))   # mismatches: reprojection gate
    o1 = rng.integers(0, 6, n).astype(np.int32)
    o2 = np.clip(o1 + np.round(np.log(np.linalg.norm(Pc1, axis=1) / np.linalg.norm(Pc2, axis=1)) / np.log(1.2)), 0, 7).astype(np.int32)
    bad_scale = rng.random(n) < 0.06
    o2[bad_scale] = np.clip(o1[bad_scale] + rng.choice([-4, 4], bad_scale.sum()), 0, 7)
    sf = SCALE_FACTORS
    kf = lambda R, t: dict(Tcw=np.hstack([R, t[:, None]]).astype(np.float32).reshape(12), K=K, scale_factors=sf,  # noqa: E731
                           level_sigma2=(sf * sf).astype(np.float32))
    return dict(kf1=kf(R1, t1), kf2=kf(R2, t2), xy1=uv1n.astype(np.float32), xy2=uv2n.astype(np.float32), octave1=o1, octave2=o2,
                Xw=Xw, clean=~(far | wrong | bad_scale), ratio_factor=float(np.float32(1.5) * np.float32(1.2)))


def make_normal_depth_case(seed, n_points=3000, max_obs=12):
    """Map points with 0..max_obs observing camera centres each for MapPoint::UpdateNormalAndDepth."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, max_obs + 1, n_points)
    counts[:5] = np.array([0, 1, 2, max_obs, 0])[:n_points][:5] if n_points >= 5 else np.array([2, 0, 1, max_obs, 0])[:n_points]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    Xw = rng.normal(0, 5.0, (n_points, 3)).astype(np.float32)
    obs_Ow = (np.repeat(Xw, counts, axis=0) + rng.normal(0, 3.0, (off[-1], 3)) + 2.0).astype(np.float32)
    ref_Ow = (Xw + rng.normal(0, 3.0, (n_points, 3)) + 1.0).astype(np.float32)
    lvl = rng.integers(0, 8, n_points)
    return dict(offsets=off, obs_Ow=obs_Ow, Xw=Xw, ref_Ow=ref_Ow, ref_level_scale=SCALE_FACTORS[lvl].astype(np.float32),
                ref_last_scale=np.full(n_points, SCALE_FACTORS[7], np.float32),
                normal=rng.normal(0, 1, (n_points, 3)).astype(np.float32), max_dist=rng.uniform(1, 9, n_points).astype(np.float32),
                min_dist=rng.uniform(0.1, 1, n_points).astype(np.float32))
