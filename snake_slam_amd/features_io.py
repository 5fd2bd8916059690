"""The `.features` cache either side of the extractor.

Reference: Snake/Preprocess/FeatureDetector.cpp:94-111 (read) and :134-139, :166-171 (write) —
``BinaryFile << std::vector<Saiga::KeyPoint<double>> << std::vector<DescriptorORB>``, files
``<tmpDir>/<frame id>.features`` and ``<frame id>_right.features``.  Saiga::BinaryFile (absent
submodule) streams a vector as its element count followed by the raw elements.  ASSUMED layout
(unverified against a file written by a real Snake-SLAM build): 64-bit little-endian count;
KeyPoint<double> = {Vec2d point; double size, angle, response; int octave} padded to 48 bytes;
DescriptorORB = 4 x u64.  Same layout as ``snake_hip::WriteFeatures`` / ``ReadFeatures``
(snake_slam_amd/cpp/snake_hip.hpp).
"""
from __future__ import annotations

import os

import numpy as np

KEYPOINT_D_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("size", "<f8"), ("angle", "<f8"), ("response", "<f8"),
                             ("octave", "<i4"), ("pad", "<i4")])
assert KEYPOINT_D_DTYPE.itemsize == 48
MAX_COUNT = 1 << 24


def feature_file(tmp_dir: str, frame_id: int, right: bool = False) -> str:
    """FeatureDetector.cpp:92-93 (frame.id + start_frame is the caller's business)."""
    return os.path.join(tmp_dir, f"{frame_id}{'_right' if right else ''}.features")


def cast_double(kps) -> np.ndarray:
    """``kp.cast<double>()`` of the extractor output (FeatureDetector.cpp:128-131)."""
    out = np.zeros(len(kps), KEYPOINT_D_DTYPE)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        out[f] = kps[f]
    return out


LAYOUT_ACK_ENV = "SNK_FEATURES_LAYOUT_ACK"


def write_features(path: str, keypoints, descriptors) -> None:
    """Writes the ASSUMED layout (module docstring).  Until a file written by a real Snake-SLAM build has been probed
    (tools/check_features_dir.py), a cache written here may be unreadable to -- or, worse, silently misread by -- the reference:
    the writer refuses unless the caller acknowledges that with SNK_FEATURES_LAYOUT_ACK=1."""
    if os.environ.get(LAYOUT_ACK_ENV) != "1":
        raise PermissionError("write_features: the .features layout is an unverified assumption (Saiga::BinaryFile is absent); set "
                              f"{LAYOUT_ACK_ENV}=1 to write it anyway, or pin the layout first with tools/check_features_dir.py")
    k = np.ascontiguousarray(keypoints, KEYPOINT_D_DTYPE)
    d = np.ascontiguousarray(descriptors, "<u8").reshape(-1, 4)
    with open(path, "wb") as f:
        f.write(np.uint64(len(k)).tobytes())
        f.write(k.tobytes())
        f.write(np.uint64(len(d)).tobytes())
        f.write(d.tobytes())


# The layouts a `BinaryFile << vector<KeyPoint<T>> << vector<DescriptorORB>` could plausibly have (saiga is absent, so the one
# written by `write_features` is an assumption): 64- or 32-bit element counts; KeyPoint<double> padded to 48 bytes, packed to
# 44, or KeyPoint<float> (24).  `probe_layout` keeps the first variant that accounts for every byte of a file.
KEYPOINT_LAYOUTS = {
    "f64x5+i32+pad (48 B)": KEYPOINT_D_DTYPE,
    "f64x5+i32 packed (44 B)": np.dtype([("x", "<f8"), ("y", "<f8"), ("size", "<f8"), ("angle", "<f8"), ("response", "<f8"), ("octave", "<i4")]),
    "f32x5+i32 (24 B)": np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")]),
}


def probe_layout(buf: bytes):
    """-> (layout name, count width in bytes, keypoints, descriptors [n, 4] uint64) for the first known layout with
    [count][nk keypoints][count][nd descriptors of 32 bytes], nk == nd, that accounts for every byte; ValueError otherwise."""
    buf = bytes(buf)
    for cw in (8, 4):
        if len(buf) < 2 * cw:
            continue
        nk = int.from_bytes(buf[:cw], "little")
        for name, dt in KEYPOINT_LAYOUTS.items():
            off = cw + nk * dt.itemsize
            if nk > MAX_COUNT or off + cw > len(buf):
                continue
            nd = int.from_bytes(buf[off:off + cw], "little")
            if nd == nk and off + cw + nd * 32 == len(buf):
                kps = np.frombuffer(buf, dt, nk, cw).copy()
                desc = np.frombuffer(buf, "<u8", nd * 4, off + cw).reshape(-1, 4).copy()
                return name, cw, kps, desc
    raise ValueError("no known layout accounts for the file's size")


def read_features_any(path: str):
    """`read_features` for a file of unknown provenance: probes the plausible layouts and returns
    (keypoints as KEYPOINT_D_DTYPE, descriptors, layout name)."""
    name, _, k, d = probe_layout(open(path, "rb").read())
    out = np.zeros(len(k), KEYPOINT_D_DTYPE)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        out[f] = k[f]
    return out, d, name


def read_features(path: str):
    """Returns (keypoints [n] KEYPOINT_D_DTYPE, descriptors [m, 4] uint64).  Raises ValueError on a
    truncated or implausible file (the reference would read garbage)."""
    buf = np.fromfile(path, np.uint8)

    def take(off, nbytes, what):
        if off + nbytes > len(buf):
            raise ValueError(f"{path}: truncated {what}")
        return buf[off:off + nbytes], off + nbytes

    raw, off = take(0, 8, "keypoint count")
    nk = int(raw.view("<u8")[0])
    if nk > MAX_COUNT:
        raise ValueError(f"{path}: bad keypoint count {nk}")
    raw, off = take(off, nk * 48, "keypoints")
    kps = raw.view(KEYPOINT_D_DTYPE).copy()
    raw, off = take(off, 8, "descriptor count")
    nd = int(raw.view("<u8")[0])
    if nd > MAX_COUNT:
        raise ValueError(f"{path}: bad descriptor count {nd}")
    raw, off = take(off, nd * 32, "descriptors")
    desc = raw.view("<u8").reshape(-1, 4).copy()
    if off != len(buf):
        raise ValueError(f"{path}: {len(buf) - off} trailing bytes")
    return kps, desc
