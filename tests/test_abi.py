"""CPU: the C-ABI library exists, loads, and exports every symbol include/snake_hip.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "snake_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"SNK_API\s+[\w\s\*]+?\b(snk_\w+)\s*\(", text)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "snk_bf_knn2" in syms and "snk_stereo_match" in syms
    assert len(syms) >= 10


def test_library_exports_every_declared_symbol():
    from snake_slam_amd import _lib

    assert _lib.LIB_PATH.exists(), "libsnake_hip.so missing: run __graft_entry__.build()"
    lib = _lib.load()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in snake_hip.h but not exported: {missing}"


def test_binding_table_covers_header():
    from snake_slam_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_strings():
    from snake_slam_amd import _lib

    lib = _lib.load()
    assert b"gfx950" in lib.snk_version()
    assert isinstance(lib.snk_last_error(), bytes)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from snake_slam_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    try:
        _lib.load()
    except _lib.SnakeHipError as e:
        assert "no CPU fallback" in str(e) or "not found" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_single_hip_runtime_in_process():
    """torch bundles its own libamdhip64; the loader must not map a second copy next to it."""
    from snake_slam_amd import _lib

    _lib.load()
    assert len(_lib.hip_runtimes_mapped()) == 1, _lib.hip_runtimes_mapped()
