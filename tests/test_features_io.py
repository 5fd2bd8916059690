"""CPU: the `.features` cache reader / writer (Python mirror and C++ adaptor agree byte for byte)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from snake_slam_amd import features_io as FIO

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(autouse=True)
def _layout_ack(monkeypatch):
    """The writer of the assumed .features layout refuses to run without this acknowledgement (features_io.write_features)."""
    monkeypatch.setenv("SNK_FEATURES_LAYOUT_ACK", "1")



def sample(n, seed=0):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, FIO.KEYPOINT_D_DTYPE)
    k["x"], k["y"] = rng.uniform(19, 733, n), rng.uniform(19, 461, n)
    k["octave"] = rng.integers(0, 4, n)
    k["size"] = 31 * 1.2 ** k["octave"]
    k["angle"] = rng.uniform(0, 360, n)
    k["response"] = rng.integers(7, 120, n)
    d = rng.integers(0, 2 ** 63, (n, 4), dtype=np.uint64)
    return k, d


@pytest.mark.parametrize("n", [0, 1, 1003])
def test_round_trip(tmp_path, n):
    k, d = sample(n)
    f = FIO.feature_file(str(tmp_path), 17, right=True)
    assert f.endswith("17_right.features")
    FIO.write_features(f, k, d)
    assert Path(f).stat().st_size == 16 + n * (48 + 32)
    k2, d2 = FIO.read_features(f)
    assert np.array_equal(k2, k) and np.array_equal(d2, d)


def test_cast_double_from_extractor_output():
    kps = np.zeros(3, [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")])
    kps["x"], kps["angle"], kps["octave"] = [1.5, 2.25, 3.0], [0.1, 359.9, 12.0], [0, 1, 3]
    k = FIO.cast_double(kps)
    assert k["x"].dtype == np.float64 and np.array_equal(k["x"], kps["x"].astype(np.float64))
    assert np.array_equal(k["angle"], kps["angle"].astype(np.float64)) and list(k["octave"]) == [0, 1, 3]


def test_truncated_and_corrupt_files_are_rejected(tmp_path):
    k, d = sample(5)
    f = str(tmp_path / "a.features")
    FIO.write_features(f, k, d)
    raw = Path(f).read_bytes()
    for cut in (4, 8 + 47, 8 + 5 * 48 + 3, len(raw) - 1):
        Path(f).write_bytes(raw[:cut])
        with pytest.raises(ValueError):
            FIO.read_features(f)
    Path(f).write_bytes(raw + b"x")
    with pytest.raises(ValueError):
        FIO.read_features(f)
    Path(f).write_bytes(np.uint64(1 << 40).tobytes() + raw[8:])
    with pytest.raises(ValueError):
        FIO.read_features(f)


CPP = r"""
#include "snake_hip.hpp"
int main(int argc, char** argv) {
    std::vector<snake_hip::KeyPointD> k; std::vector<snake_hip::DescriptorORB> d;
    snake_hip::ReadFeatures(argv[1], k, d);          // written by the Python mirror
    for (auto& kp : k) kp.response += 1.0;           // touch the data, write it back
    snake_hip::WriteFeatures(argv[2], k, d);
    // layouts of unknown provenance: the packed 44-byte / 32-bit-count variant written by the test, and the default one
    std::vector<snake_hip::KeyPointD> k2; std::vector<snake_hip::DescriptorORB> d2;
    if (snake_hip::ReadFeaturesAny(argv[4], k2, d2) != "4/44" || k2.size() != k.size() || d2 != d) return 3;
    for (size_t i = 0; i < k.size(); ++i) if (k2[i].x != k[i].x || k2[i].response + 1.0 != k[i].response || k2[i].octave != k[i].octave) return 4;
    if (snake_hip::ReadFeaturesAny(argv[1], k2, d2) != "8/48" || d2 != d) return 5;
    try { snake_hip::ReadFeaturesAny(argv[3], k2, d2); return 6; } catch (const std::runtime_error&) {}
    try { snake_hip::ReadFeatures(argv[3], k, d); return 2; } catch (const std::runtime_error&) {}  // (clobbers k / d: last)
    snk_keypoint f{1.5f, 2.5f, 31.f, 90.f, 20.f, 2};
    return snake_hip::cast_double(f).octave == 2 && argc == 5 ? 0 : 1;
}
"""


def test_cpp_adaptor_reads_and_writes_the_same_bytes(tmp_path):
    k, d = sample(64, 3)
    a, b, bad = str(tmp_path / "a.features"), str(tmp_path / "b.features"), str(tmp_path / "bad.features")
    FIO.write_features(a, k, d)
    Path(bad).write_bytes(Path(a).read_bytes()[:100])
    src = tmp_path / "fio.cpp"
    src.write_text(CPP)
    lib = ROOT / "snake_slam_amd" / "lib"
    exe = str(tmp_path / "fio")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}", str(src),
           f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    packed = str(tmp_path / "packed.features")  # 32-bit counts, KeyPoint<double> packed to 44 bytes
    kk = np.zeros(len(k), FIO.KEYPOINT_LAYOUTS["f64x5+i32 packed (44 B)"])
    for f in ("x", "y", "size", "angle", "response", "octave"):
        kk[f] = k[f]
    Path(packed).write_bytes(np.uint32(len(k)).tobytes() + kk.tobytes() + np.uint32(len(d)).tobytes() + np.ascontiguousarray(d).tobytes())
    k3, d3, name = FIO.read_features_any(packed)
    assert name.startswith("f64x5+i32 packed") and np.array_equal(k3, k) and np.array_equal(d3, d)
    r = subprocess.run([exe, a, b, bad, packed], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    k2, d2 = FIO.read_features(b)
    k["response"] += 1.0
    assert np.array_equal(k2, k) and np.array_equal(d2, d)


def test_writer_refuses_without_acknowledgement(tmp_path, monkeypatch):
    """the guessed layout must not be written silently."""
    from snake_slam_amd import features_io as FIO

    monkeypatch.delenv("SNK_FEATURES_LAYOUT_ACK", raising=False)
    k = np.zeros(3, FIO.KEYPOINT_D_DTYPE)
    d = np.zeros((3, 4), np.uint64)
    with pytest.raises(PermissionError):
        FIO.write_features(str(tmp_path / "x.features"), k, d)
    assert not (tmp_path / "x.features").exists()
