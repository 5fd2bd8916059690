"""CPU: the C-ABI library exists, loads, and exports every symbol include/snake_hip.h declares."""
import ctypes
import os
import re

import pytest
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


def test_definition_switches_abi():
    """snk_set_definition / snk_get_definition need no device: defaults are 0, unknown keys and out-of-range values are
    refused with SNK_ERR_INVALID_ARG and change nothing."""
    from snake_slam_amd import SnakeHipError, _lib

    for key, (lo, hi) in _lib.DEFINITIONS.items():
        assert _lib.get_definition(key) == 0
        _lib.set_definition(key, hi)
        assert _lib.get_definition(key) == hi
        with pytest.raises(SnakeHipError):
            _lib.set_definition(key, hi + 1)
        assert _lib.get_definition(key) == hi
        _lib.set_definition(key, 0)
    with pytest.raises(SnakeHipError):
        _lib.set_definition("no.such.key", 0)
    with pytest.raises(SnakeHipError):
        _lib.get_definition("no.such.key")


def test_dist_entry_points_fail_loudly_without_a_device():
    """snk_dist_* (RCCL result gather, include/snake_hip.h): argument checks need no device; creating a communicator without a GPU
    returns a status and an error text instead of crashing (RCCL is only opened here, by dlopen -- the library itself must load on a
    box without RCCL, which the export test above already proves for this container's CPU-only run)."""
    import ctypes as C

    import torch

    from snake_slam_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert lib.snk_dist_init(ident, 2, 2, 0, C.byref(h)) != 0 and b"rank" in lib.snk_last_error()
    assert lib.snk_dist_init(None, 0, 1, 0, C.byref(h)) != 0
    assert lib.snk_dist_init_file(b"", 0, 1, 0, 1.0, C.byref(h)) != 0
    assert lib.snk_dist_destroy(None) == 0
    if not torch.cuda.is_available():
        assert lib.snk_dist_get_unique_id(ident) != 0 or lib.snk_dist_init(ident, 0, 1, 0, C.byref(h)) != 0
        assert len(lib.snk_last_error()) > 0


def test_rccl_that_cannot_be_opened_is_an_error_code_not_a_crash(tmp_path):
    """rccl_open() on a box without a loadable RCCL (round-5 advisor: `dlerror() ? dlerror() : "?"` read the message twice, got NULL
    the second time and built a std::string from it -- every snk_dist_* entry point died with SIGSEGV).  SNK_RCCL_LIB names a file
    that is not a shared object and SNK_RCCL_STRICT forbids the fall-back to the system's copy: the header's SNK_ERR_NO_DEVICE comes
    back, with both the path and the loader's message in the error text.  In a child process: the library caches an opened RCCL."""
    import subprocess
    import sys

    bogus = tmp_path / "librccl_bogus.so.1"
    bogus.write_bytes(b"not an ELF file")
    code = (
        "import ctypes as C\n"
        "from snake_slam_amd import _lib\n"
        "lib = _lib.load()\n"
        "v = C.c_int(0)\n"
        "rc = lib.snk_dist_rccl_version(C.byref(v))\n"
        "ident = (C.c_uint8 * 128)()\n"
        "rc2 = lib.snk_dist_get_unique_id(ident)\n"
        "print(rc, rc2, lib.snk_last_error().decode())\n"
    )
    env = dict(os.environ, SNK_RCCL_LIB=str(bogus), SNK_RCCL_STRICT="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(ROOT), timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])  # -11 before the fix
    rc, rc2, text = r.stdout.strip().split(" ", 2)
    assert int(rc) == 2 and int(rc2) == 2, r.stdout  # SNK_ERR_NO_DEVICE
    assert "RCCL not found" in text and "librccl_bogus" in text and "?" not in text.split("librccl_bogus.so.1:")[1][:3], text


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/snake_hip.h must compile as C99 with no extensions (a cgo / JNI / Rust-bindgen binding
    reads it as C), and its constants must be usable in constant expressions."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text('#include "snake_hip.h"\n'
                   "typedef char id_is_128[SNK_DIST_ID_BYTES == 128 ? 1 : -1];\n"
                   "int main(void) { snk_frontend_frame f; f.capacity = 0; return (int)sizeof(snk_keypoint) == 24 && SNK_ERR_TIMEOUT == 6 && f.capacity == 0 ? 0 : 1; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(tmp_path / "abi")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(tmp_path / "abi")]).returncode == 0
