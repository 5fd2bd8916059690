"""Shared test helpers: seeded inputs and an independent numpy Hamming implementation."""
import numpy as np

SEED = 363456635  # the reference's randomSeed (reference configs/euroc.ini:3)


def rand_desc(rng, n):
    return rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)


def np_hamming_matrix(q, t):
    """Independent of the oracle: bit-unpack and count."""
    qb = np.unpackbits(np.ascontiguousarray(q).view(np.uint8).reshape(q.shape[0], 32), axis=1).astype(np.int16)
    tb = np.unpackbits(np.ascontiguousarray(t).view(np.uint8).reshape(t.shape[0], 32), axis=1).astype(np.int16)
    return (qb[:, None, :] != tb[None, :, :]).sum(axis=2).astype(np.int32)


def np_knn2(q, t):
    """Two lexicographically smallest (dist, idx) per query — the definition the scan implements."""
    nq, nt = q.shape[0], t.shape[0]
    out = np.zeros((nq, 4), np.int32)
    out[:, 0] = -1
    out[:, 1] = 256
    out[:, 2] = -1
    out[:, 3] = 256
    if nt == 0 or nq == 0:
        return out
    d = np_hamming_matrix(q, t)
    key = d.astype(np.int64) * (1 << 24) + np.arange(nt)[None, :]
    order = np.argsort(key, axis=1, kind="stable")
    out[:, 0] = order[:, 0]
    out[:, 1] = d[np.arange(nq), order[:, 0]]
    if nt > 1:
        out[:, 2] = order[:, 1]
        out[:, 3] = d[np.arange(nq), order[:, 1]]
    return out


def knn_to_array(knn):
    return np.stack([knn["idx1"], knn["dist1"], knn["idx2"], knn["dist2"]], axis=1).astype(np.int32)


def make_stereo_case(rng, nl, nr, n_levels=4, height=480, width=752, bf=47.9, dup_frac=0.6):
    """Random rectified keypoints; a fraction of the right set are noisy copies of left ones at a
    plausible disparity so that matches, ties and rejections all occur."""
    from oracle.oracle import KP64

    left = np.zeros(nl, KP64)
    left["x"] = rng.uniform(20, width - 20, nl)
    left["y"] = rng.uniform(20, height - 20, nl)
    left["angle"] = rng.uniform(0, 360, nl).astype(np.float32)
    left["octave"] = rng.integers(0, n_levels, nl)
    dl = rand_desc(rng, nl)
    right = np.zeros(nr, KP64)
    right["x"] = rng.uniform(20, width - 20, nr)
    right["y"] = rng.uniform(20, height - 20, nr)
    right["angle"] = rng.uniform(0, 360, nr).astype(np.float32)
    right["octave"] = rng.integers(0, n_levels, nr)
    dr = rand_desc(rng, nr)
    ncopy = int(min(nl, nr) * dup_frac)
    src = rng.permutation(nl)[:ncopy]
    dst = rng.permutation(nr)[:ncopy]
    for s, d in zip(src, dst):
        right["x"][d] = left["x"][s] - rng.uniform(-2, bf * 0.55)
        right["y"][d] = left["y"][s] + rng.uniform(-3, 3)
        right["angle"][d] = np.float32((left["angle"][s] + rng.uniform(-30, 30)) % 360)
        right["octave"][d] = np.clip(left["octave"][s] + rng.integers(-2, 3), 0, n_levels - 1)
        flip = rng.integers(0, 90)
        bits = rng.permutation(256)[:flip]
        desc = dl[s].copy()
        for b in bits:
            desc[b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
        dr[d] = desc
    # exact duplicates on the right to force distance ties
    for _ in range(max(1, nr // 20)):
        a, b = rng.integers(0, nr, 2)
        dr[b] = dr[a]
        right["y"][b] = right["y"][a] + rng.integers(-1, 2)
    level_scale = (np.float32(1.2) ** np.arange(n_levels)).astype(np.float32)
    return left, dl, right, dr, bf, level_scale
