"""tools/check_features_dir.py: the one-command harness that pins the extractor against `.features` files of a real Snake-SLAM
build (reference Snake/Preprocess/FeatureDetector.cpp:94-111,134-139).  CPU: layout probing, image readers, the comparison.
GPU: end to end on files written from the library's own output (100 % agreement) and on a tampered copy (reported)."""
import json
import struct
import sys
import zlib
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import check_features_dir as T  # noqa: E402


@pytest.fixture(autouse=True)
def _layout_ack(monkeypatch):
    """The writer of the assumed .features layout refuses to run without this acknowledgement (features_io.write_features)."""
    monkeypatch.setenv("SNK_FEATURES_LAYOUT_ACK", "1")



def _kps(n, rng):
    from snake_slam_amd.orb import KEYPOINT_DTYPE

    k = np.zeros(n, KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(20, 700, n).astype(np.float32), rng.uniform(20, 400, n).astype(np.float32)
    k["size"], k["angle"], k["response"] = 31.0, rng.uniform(0, 360, n).astype(np.float32), rng.integers(8, 200, n)
    k["octave"] = rng.integers(0, 4, n)
    return k, rng.integers(0, 2**64, (n, 4), dtype=np.uint64)


def test_layout_probe_accepts_the_plausible_variants(tmp_path):
    from snake_slam_amd import features_io as F

    rng = np.random.default_rng(1)
    k, d = _kps(37, rng)
    F.write_features(str(tmp_path / "a.features"), F.cast_double(k), d)
    name, cw, fk, fd = T.probe_layout((tmp_path / "a.features").read_bytes())
    assert cw == 8 and name.startswith("f64x5+i32+pad") and len(fk) == 37 and np.array_equal(fd, d)
    assert np.array_equal(fk["x"], k["x"].astype(np.float64)) and np.array_equal(fk["octave"], k["octave"])
    # 32-bit counts with packed 44-byte keypoints, and float keypoints with 64-bit counts
    for cw, lname in ((4, "f64x5+i32 packed (44 B)"), (8, "f32x5+i32 (24 B)")):
        dt = T.KP_LAYOUTS[lname]
        kk = np.zeros(37, dt)
        for f in ("x", "y", "size", "angle", "response", "octave"):
            kk[f] = k[f]
        buf = (37).to_bytes(cw, "little") + kk.tobytes() + (37).to_bytes(cw, "little") + d.tobytes()
        name, got_cw, fk, fd = T.probe_layout(buf)
        assert name == lname and got_cw == cw and np.array_equal(fd, d) and np.array_equal(fk["octave"], k["octave"])
    with pytest.raises(ValueError):
        T.probe_layout(buf[:-3])
    # empty vectors
    assert len(T.probe_layout((0).to_bytes(8, "little") * 2)[2]) == 0


def test_image_readers(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (23, 31), dtype=np.uint8)
    np.save(tmp_path / "1.npy", img)
    (tmp_path / "2.pgm").write_bytes(b"P5\n# comment\n31 23\n255\n" + img.tobytes())
    assert np.array_equal(T.read_image(str(tmp_path / "1.npy")), img)
    assert np.array_equal(T.read_image(str(tmp_path / "2.pgm")), img)

    def png(arr, filt):
        h, w = arr.shape
        rows = []
        prev = np.zeros(w, np.int32)
        for y in range(h):
            cur = arr[y].astype(np.int32)
            a = np.concatenate([[0], cur[:-1]])
            c = np.concatenate([[0], prev[:-1]])
            if filt == 0:
                line = cur
            elif filt == 1:
                line = cur - a
            elif filt == 2:
                line = cur - prev
            elif filt == 3:
                line = cur - ((a + prev) >> 1)
            else:
                pa, pb, pc = np.abs(prev - c), np.abs(a - c), np.abs(a + prev - 2 * c)
                p = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
                line = cur - p
            rows.append(bytes([filt]) + (line & 255).astype(np.uint8).tobytes())
            prev = cur

        def chunk(t, b):
            return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b""))

    for filt in range(5):
        assert np.array_equal(T._png_gray(png(img, filt)), img), filt
    (tmp_path / "3.png").write_bytes(png(img, 4))
    assert np.array_equal(T.read_image(str(tmp_path / "3.png")), img)


def test_compare_counts_what_agrees():
    from snake_slam_amd import features_io as F

    rng = np.random.default_rng(3)
    k, d = _kps(50, rng)
    fk = F.cast_double(k)
    r = T.compare(fk, d, k, d)
    assert r["same_position_octave"] == r["same_angle"] == r["same_response"] == r["identical_descriptor"] == 50 and r["same_order"]
    fk2, d2 = fk.copy(), d.copy()
    fk2["x"][0] += 1.0          # one keypoint elsewhere
    fk2["angle"][1] += 0.5      # one angle off
    d2[2, 0] ^= np.uint64(0b1011)  # three descriptor bits off
    r = T.compare(fk2, d2, k, d)
    assert r["same_position_octave"] == 49 and r["same_angle"] == 48 and r["identical_descriptor"] == 48 and not r["same_order"]
    assert r["hamming_of_matched"]["max"] == 3
    r = T.compare(fk[::-1], d[::-1], k, d)   # same set, other order
    assert r["identical_descriptor"] == 50 and not r["same_order"]


@pytest.mark.gpu
def test_end_to_end_on_the_librarys_own_files(tmp_path, capsys):
    from snake_slam_amd import features_io as F
    from snake_slam_amd import synth
    from snake_slam_amd.orb import ORBExtractor

    fdir, ldir, rdir = tmp_path / "features", tmp_path / "left", tmp_path / "right"
    for p in (fdir, ldir, rdir):
        p.mkdir()
    ext = ORBExtractor(500, 1.2, 4, 20, 7)
    for fid in (7, 8):
        l, r = synth.stereo_frame(fid, 400, 300, n_rects=100)
        np.save(ldir / f"{fid}.npy", l)
        (rdir / f"{fid}.pgm").write_bytes(b"P5\n400 300\n255\n" + r.tobytes())
        for img, right in ((l, False), (r, True)):
            k, d = ext.Detect(img)
            F.write_features(F.feature_file(str(fdir), fid, right), F.cast_double(k), d)
    ext.close()
    args = ["--features", str(fdir), "--images", str(ldir), "--right-images", str(rdir), "--nfeatures", "500", "--json", str(tmp_path / "r.json")]
    assert T.main(args) == 0
    tot = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert tot["pinned"] and tot["files"] == 4 and tot["descriptor_agreement"] == 1.0 and tot["n_file"] > 1500
    # tamper with one file: a descriptor bit and a keypoint position
    k, d = F.read_features(F.feature_file(str(fdir), 8, False))
    d[5, 1] ^= np.uint64(1)
    k["y"][9] += 2.0
    F.write_features(F.feature_file(str(fdir), 8, False), k, d)
    assert T.main(args) == 1
    tot = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert not tot["pinned"] and tot["same_position_octave"] == tot["n_file"] - 1 and tot["identical_descriptor"] == tot["n_file"] - 2
    # other extractor parameters than the files were written with: reported, not pinned
    assert T.main(args[:-2] + ["--levels", "3"]) == 1
