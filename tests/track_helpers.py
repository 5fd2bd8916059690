"""Synthetic tracking scenes for the projection matchers + independent brute-force restatements."""
import numpy as np

K_EUROC = (458.654, 457.296, 367.215, 248.375)
BF = 47.9
BOUNDS = (-12.5, -9.0, 770.25, 495.5)  # undistorted image extent (not multiples of the cell size on purpose)


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def flip_bits(rng, desc, n):
    d = desc.copy()
    for b in rng.permutation(256)[:n]:
        d[b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
    return d


def make_tracking_case(orc, rng, n_clutter=600, m_pts=500, n_levels=4, stereo_frac=0.5, taken_frac=0.05):
    """Returns (frame dict in grid order, cam, pose, level_scale, world: dict of point arrays)."""
    fx, fy, cx, cy = K_EUROC
    q = rng.normal(size=4) * 0.02 + np.array([0, 0, 0, 1.0])
    q /= np.linalg.norm(q)
    t = rng.normal(size=3) * 0.1
    pose = np.concatenate([q, t])
    R = quat_R(q)
    campos = -R.T @ t
    ls = (np.float32(1.2) ** np.arange(n_levels)).astype(np.float32)
    # world points in front of the camera (camera coordinates first)
    pc = np.stack([rng.uniform(-3, 3, m_pts), rng.uniform(-2, 2, m_pts), rng.uniform(2, 12, m_pts)], 1)
    pc[: m_pts // 20, 2] *= -1  # some behind the camera
    pw = (R.T @ (pc - t).T).T
    normal = campos - pw
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    flipn = rng.random(m_pts) < 0.1
    normal[flipn] = rng.normal(size=(int(flipn.sum()), 3))  # some with bad viewing angle
    normal[flipn] /= np.linalg.norm(normal[flipn], axis=1, keepdims=True)
    pdesc = rng.integers(0, 2**64, size=(m_pts, 4), dtype=np.uint64)
    poct = rng.integers(0, n_levels, m_pts)
    pang = rng.uniform(0, 360, m_pts).astype(np.float32)
    u = fx * pc[:, 0] / pc[:, 2] + cx
    v = fy * pc[:, 1] / pc[:, 2] + cy
    feats = []
    for i in range(m_pts):
        if pc[i, 2] <= 0 or rng.random() < 0.2:
            continue
        for _ in range(int(rng.integers(1, 3))):  # one or two nearby features (ties, second best)
            x, y = u[i] + rng.normal(0, 3.0), v[i] + rng.normal(0, 3.0)
            octv = int(np.clip(poct[i] + rng.integers(-1, 2), 0, n_levels - 1))
            d = flip_bits(rng, pdesc[i], int(rng.integers(0, 70)))
            ang = np.float32((pang[i] + rng.normal(0, 8)) % 360)
            rp = x - BF / pc[i, 2] + rng.normal(0, 0.5) if rng.random() < stereo_frac else -1.0
            feats.append((x, y, ang, octv, d, rp))
    for _ in range(n_clutter):
        feats.append((rng.uniform(BOUNDS[0], BOUNDS[2]), rng.uniform(BOUNDS[1], BOUNDS[3]), np.float32(rng.uniform(0, 360)),
                      int(rng.integers(0, n_levels)), rng.integers(0, 2**64, size=4, dtype=np.uint64),
                      rng.uniform(10, 700) if rng.random() < stereo_frac else -1.0))
    n = len(feats)
    order = rng.permutation(n)
    kps = np.zeros(n, orc.KP64)
    desc = np.zeros((n, 4), np.uint64)
    rp = np.zeros(n, np.float32)
    for j, i in enumerate(order):
        kps["x"][j], kps["y"][j], kps["angle"][j], kps["octave"][j] = feats[i][0], feats[i][1], feats[i][2], feats[i][3]
        desc[j] = feats[i][4]
        rp[j] = feats[i][5]
    perm, cell_start, cols, rows = orc.feature_grid(kps, BOUNDS)
    g_kps, g_desc, g_rp = np.zeros_like(kps), np.zeros_like(desc), np.zeros_like(rp)
    g_kps[perm], g_desc[perm], g_rp[perm] = kps, desc, rp
    taken = (rng.random(n) < taken_frac).astype(np.uint8)
    frame = dict(kps=g_kps, desc=g_desc, right_points=g_rp, taken=taken, cell_start=cell_start, bounds=BOUNDS, cols=cols,
                 rows=rows)
    world = dict(pos=pw, normal=normal, desc=pdesc, octave=poct, angle=pang, pc=pc)
    cam = (fx, fy, cx, cy, BF)
    return frame, cam, pose, ls, world, dict(kps=kps, perm=perm)


def lm_coarse(orc, world):
    m = len(world["pos"])
    pts = np.zeros(m, orc.LM_COARSE)
    pts["pos"], pts["normal"], pts["desc"], pts["octave"], pts["angle"] = world["pos"], world["normal"], world["desc"], world["octave"], world["angle"]
    return pts


def lm_fine(orc, rng, world, pose, ls):
    m = len(world["pos"])
    pts = np.zeros(m, orc.LM_FINE)
    pts["pos"], pts["normal"], pts["desc"] = world["pos"], world["normal"], world["desc"]
    R = quat_R(pose[:4])
    campos = -R.T @ pose[4:]
    dist = np.linalg.norm(world["pos"] - campos, axis=1)
    # reference depth / level such that the predicted level is near the point's octave; a few out of range
    lvl = world["octave"]
    pts["reference_scale_level"] = lvl
    pts["reference_depth"] = (dist * rng.uniform(0.8, 1.25, m)).astype(np.float32)
    far = rng.random(m) < 0.05
    pts["reference_depth"][far] *= 10
    pts["valid"] = (rng.random(m) > 0.05).astype(np.uint8)
    return pts


def hamming(a, b):
    return int(sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b)))


def brute_coarse(frame, cam, pose, pts, th, feature_error, direction, ls):
    """Independent restatement: no grid — every feature is tested against the gates, lowest index wins ties."""
    fx, fy, cx, cy, bf = cam
    R = quat_R(pose[:4])
    t = pose[4:]
    campos = -R.T @ t
    b = frame["bounds"]
    kx, ky, koct = frame["kps"]["x"], frame["kps"]["y"], frame["kps"]["octave"]
    m = len(pts)
    best = np.full(m, -1)
    bins = np.zeros(m, int)
    for i in range(m):
        pc = R @ pts["pos"][i] + t
        z = pc[2]
        if z <= 0:
            continue
        ipx, ipy = fx * pc[0] / z + cx, fy * pc[1] / z + cy
        if not (b[0] <= ipx < b[2] and b[1] <= ipy < b[3]):
            continue
        PO = campos - pts["pos"][i]
        dist = np.linalg.norm(PO)
        if PO @ pts["normal"][i] / dist < 0.5:
            continue
        lvl = int(pts["octave"][i])
        r = np.float32(np.float32(th) * ls[lvl])
        mn, mx = (lvl - 1, 100) if direction == 1 else ((0, lvl) if direction == 2 else (lvl - 1, lvl + 1))
        r2 = float(r) * float(r)
        ok = (koct >= mn) & (koct <= mx) & ((kx - ipx) ** 2 + (ky - ipy) ** 2 < r2) & (frame["taken"] == 0)
        bd, bi = 256, -1
        for j in np.nonzero(ok)[0]:
            if frame["right_points"][j] > 0 and abs((ipx - bf / z) - float(frame["right_points"][j])) > float(r) * 0.5:
                continue
            d = hamming(pts["desc"][i], frame["desc"][j])
            if d < bd:
                bd, bi = d, j
        if bd <= feature_error:
            rot = np.float32(pts["angle"][i]) - np.float32(frame["kps"]["angle"][bi])
            if rot < 0:
                rot = np.float32(rot + np.float32(360))
            bn = int(np.floor(np.float32(rot * np.float32(1.0 / 30)) + np.float32(0.5)))
            best[i], bins[i] = bi, (0 if bn == 30 else bn)
    taken = frame["taken"].copy()
    out = np.full(m, -1)
    hist = [0] * 30
    for i in range(m):
        if best[i] >= 0 and not taken[best[i]]:
            taken[best[i]] = 1
            out[i] = best[i]
            hist[bins[i]] += 1
    # three maxima
    m1 = m2 = m3 = 0
    i1 = i2 = i3 = -1
    for k, s in enumerate(hist):
        if s > m1:
            m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, k
        elif s > m2:
            m3, m2, i3, i2 = m2, s, i2, k
        elif s > m3:
            m3, i3 = s, k
    if m2 < 0.1 * m1:
        i2 = i3 = -1
    elif m3 < 0.1 * m1:
        i3 = -1
    for i in range(m):
        if out[i] >= 0 and bins[i] not in (i1, i2, i3):
            out[i] = -1
    return int((out >= 0).sum()), out
