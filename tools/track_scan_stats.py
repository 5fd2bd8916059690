"""Candidate statistics of the fine / coarse projection matchers' window scans on bench.py's tracking inputs (CPU, numpy):
    python bench.py ... --dump-track-inputs x.npz ;  python tools/track_scan_stats.py x.npz
Per wavefront of 64 consecutive local-map points: candidates per window (mean, max), grid columns per window, the trip count of the
per-lane scan (sum over the column steps of the largest column among the 64 lanes) against the balanced one (total / 64)."""
import sys

import numpy as np


def stats(name, pos, r, kps_n, cell_start, bounds, cam, cols, rows):
    fx, fy, cx, cy = cam[:4]
    ipx = fx * pos[:, 0] / pos[:, 2] + cx
    ipy = fy * pos[:, 1] / pos[:, 2] + cy
    cc = lambda p, lo, n: np.clip(np.floor((p - lo) / 20.0).astype(int), 0, n - 1)
    cx0, cx1 = cc(ipx - r, bounds[0], cols), cc(ipx + r, bounds[0], cols)
    cy0, cy1 = cc(ipy - r, bounds[1], rows), cc(ipy + r, bounds[1], rows)
    m = len(pos)
    ncol = cx1 - cx0 + 1
    maxc = int(ncol.max())
    per_col = np.zeros((m, maxc), int)
    for k in range(maxc):
        cxk = np.minimum(cx0 + k, cx1)
        cnt = cell_start[cxk * rows + cy1 + 1] - cell_start[cxk * rows + cy0]
        per_col[:, k] = np.where(cx0 + k <= cx1, cnt, 0)
    tot = per_col.sum(1)
    nw = m // 64
    pc = per_col[: nw * 64].reshape(nw, 64, maxc)
    lane_trip = pc.max(1).sum(1)           # nested loops: per column step the longest lane
    flat_trip = tot[: nw * 64].reshape(nw, 64).max(1)
    bal_trip = np.ceil(tot[: nw * 64].reshape(nw, 64).sum(1) / 64.0)
    return dict(name=name, cand_mean=float(tot.mean()), cand_max=int(tot.max()), cols_mean=float(ncol.mean()), rows_mean=float((cy1 - cy0 + 1).mean()),
                per_lane_trips=float(lane_trip.mean()), flattened_trips=float(flat_trip.mean()), balanced_rounds=float(bal_trip.mean()),
                total_per_wave=float(tot[: nw * 64].reshape(nw, 64).sum(1).mean()))


def main():
    d = np.load(sys.argv[1])
    bounds, cam, ls = d["bounds"], d["cam"], d["level_scale"]
    cols, rows = int(np.ceil((bounds[2] - bounds[0]) / 20.0)), int(np.ceil((bounds[3] - bounds[1]) / 20.0))
    out = []
    for b in range(len(d["n"])):
        cs = d["cell_start"][b]
        f, c = d["fine"][b], d["coarse"][b]
        rf = 2.5 * 4.0 * ls[np.clip(f["reference_scale_level"], 0, len(ls) - 1)]
        rc = 10.0 * ls[np.clip(c["octave"], 0, len(ls) - 1)]
        out.append((stats("fine", f["pos"], rf, d["n"][b], cs, bounds, cam, cols, rows), stats("coarse", c["pos"], rc, d["n"][b], cs, bounds, cam, cols, rows)))
    for k in (0, 1):
        keys = [x for x in out[0][k] if x != "name"]
        print(out[0][k]["name"], {x: round(float(np.mean([o[k][x] for o in out])), 2) for x in keys})


if __name__ == "__main__":
    main()
