"""Independent restatement of the quadtree distribution in its sorted-key ("Morton") form — the
formulation the HIP kernel uses — to cross-check the oracle's explicit node-list version."""
import numpy as np

MIN_BORDER = 16


def n_roots(W, H):
    return max(1, min(255, (2 * W + H) // (2 * H)))


def point_key(x, y, W, H):
    nI = n_roots(W, H)
    root = (x * nI) // W
    x0, x1 = (root * W + nI - 1) // nI, ((root + 1) * W + nI - 1) // nI
    y0, y1 = 0, H
    key = root
    for _ in range(16):
        mx, my = x0 + (x1 - x0 + 1) // 2, y0 + (y1 - y0 + 1) // 2
        c = (1 if x >= mx else 0) + (2 if y >= my else 0)
        if x >= mx:
            x0 = mx
        else:
            x1 = mx
        if y >= my:
            y0 = my
        else:
            y1 = my
        key = (key << 2) | c
    return key


def distribute(xs, ys, scores, w, h, N):
    """xs, ys absolute level coordinates.  Returns indices (into the input) in output order."""
    n = len(xs)
    if n == 0 or N <= 0:
        return []
    W, H = w - 2 * MIN_BORDER, h - 2 * MIN_BORDER
    keys = [point_key(int(xs[i]) - MIN_BORDER, int(ys[i]) - MIN_BORDER, W, H) for i in range(n)]
    order = sorted(range(n), key=lambda i: keys[i])
    k = [keys[i] for i in order]
    pre = lambda i, d: k[i] >> (2 * (16 - d))

    def segments(d, lo=0, hi=None):
        hi = n if hi is None else hi
        segs, s = [], lo
        for i in range(lo + 1, hi + 1):
            if i == hi or pre(i, d) != pre(s, d):
                segs.append((s, i))
                s = i
        return segs

    fd = [0] * n  # final depth per sorted point
    d = 0
    size_prev = len(segments(0))
    finished = False
    while not finished:
        d += 1
        if d > 16:
            d = 16
            break
        segs = segments(d)
        size = len(segs)
        n_exp = sum(1 for a, b in segs if b - a > 1)
        if size >= N or size == size_prev:
            finished = True
        elif size + 3 * n_exp > N:
            for i in range(n):
                fd[i] = d
            lst = [(a, b, d) for a, b in segs if b - a > 1]
            while True:
                prev = size
                lst.sort(key=lambda t: (-(t[1] - t[0]), pre(t[0], t[2])))
                nxt = []
                for a, b, dd in lst:
                    ch = segments(dd + 1, a, b) if dd < 16 else [(a, b)]
                    size += len(ch) - 1
                    for i in range(a, b):
                        fd[i] = min(16, dd + 1)
                    nxt += [(ca, cb, dd + 1) for ca, cb in ch if cb - ca > 1 and dd + 1 <= 16]
                    if size >= N:
                        break
                if size >= N or size == prev:
                    break
                lst = nxt
            d = None
            break
        size_prev = size
    if d is not None:
        for i in range(n):
            fd[i] = d
    out, s = [], 0
    for i in range(1, n + 1):
        if i == n or pre(i, fd[i]) != pre(i - 1, fd[i]):
            best = min(range(s, i), key=lambda j: (-int(scores[order[j]]), int(ys[order[j]]), int(xs[order[j]])))
            out.append(order[best])
            s = i
    return out
