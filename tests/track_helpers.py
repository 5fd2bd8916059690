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


# ------------------------------------------------------------------ local-mapping matchers -----
def fusion_points(orc, rng, world, pose, ls):
    m = len(world["pos"])
    pts = np.zeros(m, orc.FUSION_POINT)
    pts["pos"], pts["normal"], pts["desc"] = world["pos"], world["normal"], world["desc"]
    R = quat_R(pose[:4])
    campos = -R.T @ pose[4:]
    dist = np.linalg.norm(world["pos"] - campos, axis=1)
    pts["reference_scale_level"] = world["octave"]
    pts["reference_depth"] = (dist * rng.uniform(0.8, 1.25, m)).astype(np.float32)
    far = rng.random(m) < 0.05
    pts["reference_depth"][far] *= 10
    pts["observations"] = rng.integers(1, 6, m)
    pts["id"] = rng.permutation(10 * m)[:m]
    return pts


def _scale_prediction(orc, ref_depth, ref_level, dist, ls):
    log_f = orc.det_log(float(ls[1]) / float(ls[0]))
    p = float(ref_level) + orc.det_log(float(ref_depth) / dist) / log_f
    return min(max(p, 0.0), float(len(ls) - 1))


def brute_fuse(orc, frame, cam, pose, pts, mask, th, obs_factor, feature_th, ls):
    """Independent restatement without the grid: every feature is tested, lowest index wins ties."""
    fx, fy, cx, cy, bf = cam
    R, t = quat_R(pose[:4]), pose[4:]
    campos = -R.T @ t
    b = frame["bounds"]
    kx, ky, koct, rp = frame["kps"]["x"], frame["kps"]["y"], frame["kps"]["octave"], frame["right_points"]
    best = np.full(len(pts), -1)
    th2 = np.float32(th) * np.float32(th)
    for i in range(len(pts)):
        if mask is not None and not mask[i]:
            continue
        pc = R @ pts["pos"][i] + t
        if pc[2] <= 0:
            continue
        ipx, ipy = fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy
        if not (b[0] <= ipx < b[2] and b[1] <= ipy < b[3]):
            continue
        PO = campos - pts["pos"][i]
        dist = np.linalg.norm(PO)
        lvl = int(np.clip(pts["reference_scale_level"][i], 0, len(ls) - 1))
        rd = float(pts["reference_depth"][i])
        if dist < 0.8 * rd * float(ls[lvl]) / float(ls[-1]) or dist > 1.2 * rd * float(ls[lvl]):
            continue
        if PO @ pts["normal"][i] < 0.5 * dist:
            continue
        of = np.float32(obs_factor) if pts["observations"][i] <= 2 else np.float32(1.0)
        radius = float(np.float32(of * np.float32(th)))
        gate = float(np.float32(th2 * of))
        pred = _scale_prediction(orc, rd, pts["reference_scale_level"][i], dist, ls)
        e2 = (kx - ipx) ** 2 + (ky - ipy) ** 2
        ok = (e2 < radius * radius) & (np.abs(pred - koct) <= 1.0)
        e2 = e2 + np.where(rp > 0, ((ipx - bf / pc[2]) - rp.astype(np.float64)) ** 2, 0.0)
        ok &= ~(e2 > gate)
        bd, bi = 256, -1
        for j in np.nonzero(ok)[0]:
            d = hamming(pts["desc"][i], frame["desc"][j])
            if d < bd:
                bd, bi = d, int(j)
        if bi >= 0 and bd <= feature_th:
            best[i] = bi
    return int((best >= 0).sum()), best


def essential(pose1, pose2):
    """E with x2^T E x1 = 0 for normalized points of the two world->camera poses."""
    R1, R2 = quat_R(pose1[:4]), quat_R(pose2[:4])
    R21 = R2 @ R1.T
    t21 = pose2[4:] - R21 @ pose1[4:]
    tx = np.array([[0, -t21[2], t21[1]], [t21[2], 0, -t21[0]], [-t21[1], t21[0], 0]])
    return tx @ R21


def make_triangulation_case(orc, rng, m_pts=500, n_clutter=300, n_levels=4):
    """Two keyframes looking at the same points.  Returns a dict with everything the matcher needs."""
    fx, fy, cx, cy = K_EUROC
    poses = []
    for k in range(2):
        q = rng.normal(size=4) * 0.02 + np.array([0, 0, 0, 1.0])
        q /= np.linalg.norm(q)
        poses.append(np.concatenate([q, rng.normal(size=3) * 0.05 + np.array([0.6 * k, 0, 0])]))
    R1, t1 = quat_R(poses[0][:4]), poses[0][4:]
    R2, t2 = quat_R(poses[1][:4]), poses[1][4:]
    pc1 = np.stack([rng.uniform(-3, 3, m_pts), rng.uniform(-2, 2, m_pts), rng.uniform(3, 9, m_pts)], 1)
    pw = (pc1 - t1) @ R1
    pc2 = pw @ R2.T + t2
    pdesc = rng.integers(0, 2**64, size=(m_pts, 4), dtype=np.uint64)

    def features(pc, noise):
        u = fx * pc[:, 0] / pc[:, 2] + cx + rng.normal(0, noise, len(pc))
        v = fy * pc[:, 1] / pc[:, 2] + cy + rng.normal(0, noise, len(pc))
        d = np.stack([flip_bits(rng, pdesc[i], int(rng.integers(0, 60))) for i in range(len(pc))])
        cu, cv = rng.uniform(BOUNDS[0], BOUNDS[2], n_clutter), rng.uniform(BOUNDS[1], BOUNDS[3], n_clutter)
        cd = rng.integers(0, 2**64, size=(n_clutter, 4), dtype=np.uint64)
        k = np.zeros(len(pc) + n_clutter, orc.KP64)
        k["x"], k["y"] = np.concatenate([u, cu]), np.concatenate([v, cv])
        k["octave"] = rng.integers(0, n_levels, len(k))
        k["angle"] = rng.uniform(0, 360, len(k)).astype(np.float32)
        return k, np.concatenate([d, cd])

    k1, d1 = features(pc1, 0.3)
    k2, d2 = features(pc2, 0.3)
    inb = lambda k: (k["x"] >= BOUNDS[0]) & (k["x"] < BOUNDS[2]) & (k["y"] >= BOUNDS[1]) & (k["y"] < BOUNDS[3])
    keep1, keep2 = inb(k1), inb(k2)
    k1, d1, k2, d2 = k1[keep1], d1[keep1], k2[keep2], d2[keep2]
    o1, o2 = rng.permutation(len(k1)), rng.permutation(len(k2))
    k1, d1, k2, d2 = k1[o1], d1[o1], k2[o2], d2[o2]
    perm, cell_start, cols, rows = orc.feature_grid(k2, BOUNDS)
    g_k2, g_d2 = np.zeros_like(k2), np.zeros_like(d2)
    g_k2[perm], g_d2[perm] = k2, d2
    np1 = np.stack([(k1["x"] - cx) / fx, (k1["y"] - cy) / fy], 1)
    np2 = np.stack([(g_k2["x"] - cx) / fx, (g_k2["y"] - cy) / fy], 1)
    has1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    has2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
    frame2 = dict(kps=g_k2, desc=g_d2, right_points=np.full(len(k2), -1, np.float32), taken=has2, cell_start=cell_start,
                  bounds=BOUNDS, cols=cols, rows=rows)
    # coarse depth grid: one value per 4 x 4 cells (80 px), near the true depth range, some far off
    grid = rng.uniform(4.0, 8.0, ((rows + 3) // 4, (cols + 3) // 4))
    grid[rng.random(grid.shape) < 0.1] = 40.0
    return dict(grid=grid, pose1=poses[0], pose2=poses[1], cam=(fx, fy, cx, cy, BF), kps1=k1, np1=np1, desc1=d1, has1=has1,
                frame2=frame2, np2=np2, E=essential(poses[0], poses[1]))


def brute_triangulation(case, epipolar_distance, feature_distance):
    fx, fy, cx, cy, bf = case["cam"]
    R1, t1 = quat_R(case["pose1"][:4]), case["pose1"][4:]
    R2, t2 = quat_R(case["pose2"][:4]), case["pose2"][4:]
    f2 = case["frame2"]
    b = f2["bounds"]
    kx, ky = f2["kps"]["x"], f2["kps"]["y"]
    cols, rows = f2["cols"], f2["rows"]
    th2 = (np.float64(np.float32(epipolar_distance)) / fx) ** 2
    E = case["E"]
    out = np.full(len(case["kps1"]), -1)
    for i in range(len(case["kps1"])):
        if case["has1"][i]:
            continue
        x, y = case["kps1"]["x"][i], case["kps1"]["y"][i]
        cxi = int(np.clip(np.floor((x - b[0]) / 20.0), 0, cols - 1))
        cyi = int(np.clip(np.floor((y - b[1]) / 20.0), 0, rows - 1))
        z = case["grid"][cyi // 4, cxi // 4]
        pc = np.array([(x - cx) / fx * z, (y - cy) / fy * z, z])
        wp = R1.T @ (pc - t1)
        p2 = R2 @ wp + t2
        ipx, ipy = fx * p2[0] / p2[2] + cx, fy * p2[1] / p2[2] + cy
        if not (b[0] <= ipx < b[2] and b[1] <= ipy < b[3]):
            continue
        l = E @ np.array([case["np1"][i][0], case["np1"][i][1], 1.0])
        d = case["np2"] @ l[:2] + l[2]
        ok = ((kx - ipx) ** 2 + (ky - ipy) ** 2 < 400.0) & (f2["taken"] == 0) & ~(d * d / (l[0] ** 2 + l[1] ** 2) > th2)
        bd, bi = 50, -1
        for j in np.nonzero(ok)[0]:
            h = hamming(case["desc1"][i], f2["desc"][j])
            if h > feature_distance or h > bd:
                continue
            bd, bi = h, int(j)
        out[i] = bi
    return int((out >= 0).sum()), out


# ---- bag-of-words / brute-force triangulation matchers (MappingORBMatcher::SearchForTriangulation2 / BF) ----
def make_bow(labels):
    """Flatten 'feature i belongs to vocabulary node labels[i]' into (node_id ascending, node_start, features), the
    order DBoW2 fills a FeatureVector in (features of a node in ascending index)."""
    labels = np.asarray(labels)
    ids = np.unique(labels)
    start, feat = [0], []
    for nid in ids:
        members = np.nonzero(labels == nid)[0]
        feat.extend(members.tolist())
        start.append(len(feat))
    return ids.astype(np.uint32), np.asarray(start, np.int32), np.asarray(feat, np.int32)


def make_bow_case(rng, m_pts=400, n_clutter=200, n_nodes=60, max_flip=60, has_frac=0.3):
    """Two keyframes seeing the same points; node labels agree for most true correspondences."""
    fx, fy, cx, cy = K_EUROC
    poses = []
    for k in range(2):
        q = rng.normal(size=4) * 0.02 + np.array([0, 0, 0, 1.0])
        q /= np.linalg.norm(q)
        poses.append(np.concatenate([q, rng.normal(size=3) * 0.05 + np.array([0.6 * k, 0, 0])]))
    R1, t1 = quat_R(poses[0][:4]), poses[0][4:]
    R2, t2 = quat_R(poses[1][:4]), poses[1][4:]
    pc1 = np.stack([rng.uniform(-3, 3, m_pts), rng.uniform(-2, 2, m_pts), rng.uniform(3, 9, m_pts)], 1)
    pw = (pc1 - t1) @ R1
    pc2 = pw @ R2.T + t2
    pdesc = rng.integers(0, 2**64, size=(m_pts, 4), dtype=np.uint64)
    plabel = rng.integers(0, n_nodes, m_pts) * 7 + 3  # sparse, non-contiguous node ids

    def view(pc):
        n = len(pc) + n_clutter
        xy = np.zeros((n, 2))
        xy[: len(pc), 0] = pc[:, 0] / pc[:, 2] + rng.normal(0, 0.3 / fx, len(pc))
        xy[: len(pc), 1] = pc[:, 1] / pc[:, 2] + rng.normal(0, 0.3 / fy, len(pc))
        xy[len(pc):] = rng.uniform(-0.7, 0.7, (n_clutter, 2))
        d = np.concatenate([np.stack([flip_bits(rng, pdesc[i], int(rng.integers(0, max_flip + 1))) for i in range(len(pc))])
                            if len(pc) else np.zeros((0, 4), np.uint64),
                            rng.integers(0, 2**64, size=(n_clutter, 4), dtype=np.uint64)])
        lab = np.concatenate([np.where(rng.random(len(pc)) < 0.9, plabel, rng.integers(0, n_nodes, len(pc)) * 7 + 3),
                              rng.integers(0, n_nodes + 10, n_clutter) * 7 + 3])
        o = rng.permutation(n)
        return xy[o], d[o], lab[o], (rng.random(n) < has_frac).astype(np.uint8)

    np1, d1, l1, h1 = view(pc1)
    np2, d2, l2, h2 = view(pc2)
    return dict(cam=(fx, fy, cx, cy, BF), E=essential(poses[0], poses[1]), np1=np1, desc1=d1, has1=h1, bow1=make_bow(l1),
                np2=np2, desc2=d2, has2=h2, bow2=make_bow(l2))


def _epi2(E, p1, p2):
    l = E @ np.array([p1[0], p1[1], 1.0])
    d = p2[0] * l[0] + p2[1] * l[1] + l[2]
    return d * d / (l[0] * l[0] + l[1] * l[1])


def _ham(a, b):
    return sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b))


def brute_triangulation_bow(c, epipolar_distance, feature_distance):
    """numpy / python restatement: for every common node, every unmatched feature of keyframe 1 takes the LAST feature
    of minimal Hamming distance (<= min(feature_distance, 50)) among the node's unmatched features of keyframe 2 that
    lie strictly inside the epipolar band."""
    fx = c["cam"][0]
    th2 = (np.float64(np.float32(epipolar_distance) * np.float32(2)) / fx) ** 2
    ids1, s1, f1 = c["bow1"]
    ids2, s2, f2 = c["bow2"]
    pos2 = {int(n): k for k, n in enumerate(ids2)}
    pairs = []
    for a, nid in enumerate(ids1):
        if int(nid) not in pos2:
            continue
        b = pos2[int(nid)]
        for i in f1[s1[a]:s1[a + 1]]:
            if c["has1"][i]:
                continue
            best, bd = -1, 10**9
            for j in f2[s2[b]:s2[b + 1]]:
                if c["has2"][j]:
                    continue
                d = _ham(c["desc1"][i], c["desc2"][j])
                if d > feature_distance or d > 50 or not (_epi2(c["E"], c["np1"][i], c["np2"][j]) < th2):
                    continue
                if d <= bd:
                    best, bd = int(j), d
            if best >= 0:
                pairs.append((int(i), best))
    return pairs


def brute_triangulation_bf(c, feature_distance):
    fx = c["cam"][0]
    th2 = (10 / fx) ** 2
    out = np.full(len(c["np1"]), -1)
    E = c["E"]
    d1 = c["desc1"]
    d2 = c["desc2"]
    x2 = d2[None, :, :]
    for i in range(len(out)):
        if c["has1"][i]:
            continue
        l = E @ np.array([c["np1"][i][0], c["np1"][i][1], 1.0])
        dd = c["np2"][:, 0] * l[0] + c["np2"][:, 1] * l[1] + l[2]
        ok = ~(dd * dd / (l[0] * l[0] + l[1] * l[1]) > th2) & (c["has2"] == 0)
        if not ok.any():
            continue
        x = d2 ^ d1[i][None, :]
        ham = np.zeros(len(d2), np.int64)
        for w in range(4):
            ham += np.array([bin(int(v)).count("1") for v in x[:, w]])
        ok &= (ham <= feature_distance) & (ham <= 50)
        if ok.any():
            m = ham[ok].min()
            out[i] = np.nonzero(ok & (ham == m))[0][-1]
    return out


# ---- DeferredMapper::Relink (per-observation search) ----
def make_relink_case(orc, rng, n_base=500, twin_frac=0.5, point_frac=0.8):
    """A keyframe whose features hold map points, with 'twin' features within a pixel of many projections.
    Returns (frame dict in grid order, cam, pose, queries)."""
    fx, fy, cx, cy = K_EUROC
    q = rng.normal(size=4) * 0.05 + np.array([0, 0, 0, 1.0])
    q /= np.linalg.norm(q)
    pose = np.concatenate([q, rng.normal(size=3) * 0.2])
    R, t = quat_R(q), pose[4:]
    bx = rng.uniform(BOUNDS[0] + 5, BOUNDS[2] - 5, n_base)
    by = rng.uniform(BOUNDS[1] + 5, BOUNDS[3] - 5, n_base)
    feats, queries = [], []  # feats: (x, y, desc, right)
    for i in range(n_base):
        dmp = rng.integers(0, 2**64, size=4, dtype=np.uint64)
        z = rng.uniform(3, 9)
        # projection of the point: near the feature, sometimes an outlier, sometimes behind the camera
        off = rng.normal(0, 0.7, 2) if rng.random() < 0.85 else rng.normal(0, 3.0, 2)
        ipx, ipy = bx[i] + off[0], by[i] + off[1]
        kd = flip_bits(rng, dmp, int(rng.choice([0, 0, 3, 10, 20, 30, 40])))
        right = -1.0 if rng.random() < 0.5 else ipx - BF / z + rng.normal(0, 0.5)
        me = len(feats)
        feats.append((bx[i], by[i], kd, right))
        if rng.random() < twin_frac:
            for _ in range(int(rng.integers(1, 4))):
                d = rng.normal(0, 0.45, 2)
                td = flip_bits(rng, dmp, int(rng.choice([0, 2, 5, 10, 20, 24, 25, 30])))
                tr = -1.0 if rng.random() < 0.5 else ipx - BF / z + rng.normal(0, 0.8)
                feats.append((ipx + d[0], ipy + d[1], td, tr))
        if rng.random() < point_frac:
            if rng.random() < 0.03:
                z = -z
            pc = np.array([(ipx - cx) / fx * z, (ipy - cy) / fy * z, z])
            wp = R.T @ (pc - t)
            has_alt = int(rng.random() < 0.6)
            alt = flip_bits(rng, dmp, int(rng.integers(0, 40)))
            queries.append((wp, dmp, alt, me, has_alt))
    n = len(feats)
    k = np.zeros(n, orc.KP64)
    k["x"] = np.clip([f[0] for f in feats], BOUNDS[0], BOUNDS[2] - 1e-6)
    k["y"] = np.clip([f[1] for f in feats], BOUNDS[1], BOUNDS[3] - 1e-6)
    k["octave"] = rng.integers(0, 4, n)
    d = np.stack([f[2] for f in feats])
    rp = np.array([f[3] for f in feats], np.float32)
    perm, cell_start, cols, rows = orc.feature_grid(k, BOUNDS)
    g_k, g_d, g_r = np.zeros_like(k), np.zeros_like(d), np.zeros_like(rp)
    g_k[perm], g_d[perm], g_r[perm] = k, d, rp
    frame = dict(kps=g_k, desc=g_d, right_points=g_r, taken=np.zeros(n, np.uint8), cell_start=cell_start, bounds=BOUNDS,
                 cols=cols, rows=rows)
    qs = np.zeros(len(queries), orc.RELINK_QUERY)
    for j, (wp, dmp, alt, me, has_alt) in enumerate(queries):
        qs["pos"][j], qs["desc"][j], qs["alt_desc"][j] = wp, dmp, alt
        qs["feature"][j], qs["has_alt"][j] = perm[me], has_alt
    return frame, (fx, fy, cx, cy, BF), pose, qs


def brute_relink(frame, cam, pose, qs, radius=0.8, outlier_threshold=2.1, feature_threshold=25):
    """Exhaustive restatement: scan EVERY feature of the keyframe instead of the grid cells."""
    fx, fy, cx, cy, bf = cam
    R, t = quat_R(pose[:4]), pose[4:]
    kx, ky, rp = frame["kps"]["x"], frame["kps"]["y"], frame["right_points"]
    r2 = np.float64(np.float32(radius) * np.float32(radius))
    action, best = np.zeros(len(qs), np.int32), np.full(len(qs), -1, np.int32)
    for n in range(len(qs)):
        i = int(qs["feature"][n])
        pc = R @ qs["pos"][n] + t
        z = pc[2]
        if z <= 0:
            action[n] = 1
            continue
        ipx, ipy = fx * pc[0] / z + cx, fy * pc[1] / z + cy
        rep2 = (ipx - kx[i]) ** 2 + (ipy - ky[i]) ** 2
        if rep2 > outlier_threshold * outlier_threshold:
            action[n] = 1
            continue
        fd = _ham(qs["desc"][n], frame["desc"][i])
        if fd == 0 and qs["has_alt"][n]:
            fd = _ham(qs["desc"][n], qs["alt_desc"][n])
        e2 = (kx - ipx) ** 2 + (ky - ipy) ** 2
        cand = np.nonzero(e2 < r2)[0]
        bd, bi = fd, -1
        # grid iteration order == ascending index in grid order (cells x-major, then y, then members)
        for j in cand:
            if j == i or (ipx - kx[j]) ** 2 + (ipy - ky[j]) ** 2 > rep2:
                continue
            if rp[j] > 0:
                er = (ipx - bf / z) - np.float64(rp[j])
                if er * er > rep2 * 2.0:
                    continue
            d2 = _ham(qs["desc"][n], frame["desc"][j])
            if d2 < feature_threshold and d2 < bd:
                bd, bi = d2, int(j)
        if bi >= 0:
            action[n], best[n] = 2, bi
    return action, best
