"""Builds tools/cpp/frontend_latency.cpp (g++, the C++ adaptor header, no HIP in the translation unit), writes the stereo pairs it reads
and runs it: the reference's per-frame front-end without Python in the timed region (snk_frontend_process against the six per-seam calls).
    python tools/frontend_latency_cpp.py [n_pairs] [calls]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snake_slam_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    w, h = 752, 480
    lib = os.path.join(ROOT, "snake_slam_amd", "lib")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "frontend_latency")
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", f"-I{ROOT}/include", f"-I{ROOT}/snake_slam_amd/cpp", f"{ROOT}/tools/cpp/frontend_latency.cpp",
               f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
        subprocess.run(cmd, check=True)
        for k in range(n):
            l, r = synth.stereo_frame(300 + k, w, h)
            np.ascontiguousarray(l, np.uint8).tofile(os.path.join(d, f"pair{k}_left.bin"))
            np.ascontiguousarray(r, np.uint8).tofile(os.path.join(d, f"pair{k}_right.bin"))
        r = subprocess.run([exe, d, str(n), str(w), str(h), str(calls)], capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        sys.stderr.write(r.stderr[-2000:])
        return r.returncode


if __name__ == "__main__":
    sys.exit(main())
