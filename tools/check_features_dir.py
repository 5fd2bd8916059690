#!/usr/bin/env python3
"""Pin the extractor against a REAL Snake-SLAM build: compare `<id>.features` files with this library's output.

The reference caches what its extractor produced per frame (`fd_bufferToFile`, Snake/Preprocess/FeatureDetector.cpp:94-111 read,
:134-139 / :166-171 write): `<tmpDir>/<frame id>.features` and `<frame id>_right.features`, each
`BinaryFile << vector<KeyPoint<double>> << vector<DescriptorORB>`.  Nobody can build saiga in this repository's container, so
reference parity of the extractor is unpinned; whoever HAS a Snake-SLAM build runs it once with `fd_bufferToFile`,
copies the files and the images next to each other and runs

    python tools/check_features_dir.py --features DIR --images DIR [--image-pattern "{id}.png"] \
        [--nfeatures 1000 --levels 4 --scale 1.2 --ini-th 20 --min-th 7] [--json report.json]

The tool extracts every image with the library (GPU, the product path), reads the matching `.features` file and reports, per
file and in total: keypoint counts, how many of the file's keypoints this library reproduces at exactly the same position and
octave, and of those how many have the same angle / response and a bit-identical descriptor (plus the Hamming distances of the
rest).  100 % everywhere pins "snk-orb v1" to saiga; anything else says which stage to look at first (positions -> pyramid /
FAST / distribution; angle -> orientation; descriptors -> blur / pattern / rounding).

File layout: saiga's BinaryFile is absent, so the layout of snake_slam_amd/features_io.py is an assumption.  `probe_layout`
therefore tries the plausible variants (32 / 64-bit counts; KeyPoint<double> of 48 bytes, packed 44, or KeyPoint<float> of 24)
and keeps the one that accounts for every byte of the file; the report says which one matched.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from snake_slam_amd.features_io import KEYPOINT_LAYOUTS as KP_LAYOUTS, probe_layout  # noqa: E402  (the probing lives with the format)


def _png_gray(data: bytes) -> np.ndarray:
    """Minimal PNG decoder: 8-bit grayscale (colour type 0) or RGB / RGBA (2 / 6, converted with the BT.601 weights),
    non-interlaced.  Used when PIL is absent."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 6):
        raise ValueError("PNG: only 8-bit non-interlaced gray / RGB / RGBA")
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(w * ch, np.int32)
        if f in (0, 2):
            cur = (line + (prev if f == 2 else 0)) & 255
        else:
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b, c = prev[x], (prev[x - ch] if x >= ch else 0)
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y], prev = cur, cur
    if ch == 1:
        return out
    rgb = out.reshape(h, w, ch)[:, :, :3].astype(np.float64)
    return np.clip(np.rint(rgb @ np.array([0.299, 0.587, 0.114])), 0, 255).astype(np.uint8)


def read_image(path: str) -> np.ndarray:
    """8-bit gray image from .npy, binary PGM (P5), or PNG / anything PIL reads."""
    if path.endswith(".npy"):
        a = np.load(path)
    elif path.lower().endswith(".pgm"):
        raw = open(path, "rb").read()
        m = re.match(rb"P5\s+(?:#[^\n]*\n\s*)*(\d+)\s+(\d+)\s+(\d+)\s", raw)
        if not m or int(m.group(3)) != 255:
            raise ValueError(f"{path}: not an 8-bit binary PGM")
        w, h = int(m.group(1)), int(m.group(2))
        a = np.frombuffer(raw, np.uint8, w * h, m.end()).reshape(h, w)
    else:
        try:
            from PIL import Image

            a = np.asarray(Image.open(path).convert("L"))
        except ImportError:
            a = _png_gray(open(path, "rb").read())
    a = np.ascontiguousarray(a)
    if a.ndim != 2 or a.dtype != np.uint8:
        raise ValueError(f"{path}: need an 8-bit single-channel image, got {a.dtype} {a.shape}")
    return a


def compare(file_kps, file_desc, kps, desc) -> dict:
    """Agreement of the file's features with the library's (`kps` KEYPOINT_DTYPE float32, `desc` [n, 4] uint64)."""
    def key(x, y, o):
        return (int(np.rint(float(x) * 64)), int(np.rint(float(y) * 64)), int(o))  # level-0 coordinates are float32: compare on a 1/64 px grid

    ours = {key(k["x"], k["y"], k["octave"]): i for i, k in enumerate(kps)}
    same_pos = same_angle = same_resp = same_desc = 0
    ham = []
    for j in range(len(file_kps)):
        fk = file_kps[j]
        i = ours.get(key(fk["x"], fk["y"], fk["octave"]))
        if i is None:
            continue
        same_pos += 1
        same_angle += int(abs(float(fk["angle"]) - float(kps["angle"][i])) <= 1e-4)
        same_resp += int(abs(float(fk["response"]) - float(kps["response"][i])) <= 1e-4)
        d = int(sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(file_desc[j], desc[i])))
        same_desc += int(d == 0)
        ham.append(d)
    order_same = bool(len(file_kps) == len(kps) and same_pos == len(kps) and
                      all(ours.get(key(fk["x"], fk["y"], fk["octave"])) == j for j, fk in enumerate(file_kps)))
    return {"n_file": int(len(file_kps)), "n_library": int(len(kps)), "same_position_octave": same_pos, "same_angle": same_angle,
            "same_response": same_resp, "identical_descriptor": same_desc, "same_order": order_same,
            "hamming_of_matched": {"mean": float(np.mean(ham)) if ham else None, "max": int(max(ham)) if ham else None}}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--features", required=True, help="directory with <id>.features (and <id>_right.features)")
    ap.add_argument("--images", required=True, help="directory with the left images")
    ap.add_argument("--right-images", default=None, help="directory with the right images (for *_right.features)")
    ap.add_argument("--image-pattern", default="{id}.png", help="image file name for frame id (also tried: .pgm, .npy)")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.2)
    ap.add_argument("--ini-th", type=int, default=20)
    ap.add_argument("--min-th", type=int, default=7)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--json", default=None, help="write the full report here")
    a = ap.parse_args(argv)

    from snake_slam_amd.orb import ORBExtractor

    ext = ORBExtractor(a.nfeatures, a.scale, a.levels, a.ini_th, a.min_th, device=a.device)
    files = sorted(f for f in os.listdir(a.features) if f.endswith(".features"))
    report, tot = [], {"files": 0, "n_file": 0, "same_position_octave": 0, "same_angle": 0, "same_response": 0, "identical_descriptor": 0,
                       "same_order_files": 0, "skipped": 0}
    for f in files:
        m = re.match(r"^(\d+)(_right)?\.features$", f)
        if not m:
            continue
        fid, right = m.group(1), bool(m.group(2))
        img_dir = a.right_images if right else a.images
        img_path = None
        if img_dir:
            for pat in (a.image_pattern, "{id}.pgm", "{id}.npy"):
                p = os.path.join(img_dir, pat.format(id=fid))
                if os.path.exists(p):
                    img_path = p
                    break
        if img_path is None:
            tot["skipped"] += 1
            report.append({"file": f, "error": "no image"})
            continue
        try:
            layout, cw, fk, fd = probe_layout(open(os.path.join(a.features, f), "rb").read())
            kps, desc = ext.Detect(read_image(img_path))
            r = dict(file=f, image=os.path.basename(img_path), layout=layout, count_bytes=cw, **compare(fk, fd, kps, desc))
        except (ValueError, AssertionError) as e:
            tot["skipped"] += 1
            report.append({"file": f, "error": str(e)})
            continue
        report.append(r)
        tot["files"] += 1
        for k in ("n_file", "same_position_octave", "same_angle", "same_response", "identical_descriptor"):
            tot[k] += r[k]
        tot["same_order_files"] += int(r["same_order"])
    ext.close()
    n = max(1, tot["n_file"])
    tot["position_agreement"] = round(tot["same_position_octave"] / n, 6)
    tot["descriptor_agreement"] = round(tot["identical_descriptor"] / n, 6)
    tot["pinned"] = bool(tot["files"] > 0 and tot["same_order_files"] == tot["files"] and tot["identical_descriptor"] == tot["n_file"]
                         and tot["same_angle"] == tot["n_file"] and tot["same_response"] == tot["n_file"])
    out = {"total": tot, "files": report}
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps(tot))
    return 0 if tot["pinned"] else 1


if __name__ == "__main__":
    sys.exit(main())
