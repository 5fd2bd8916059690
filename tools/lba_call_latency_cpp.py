"""Builds tools/cpp/lba_call_latency.cpp (g++, the C++ adaptor header, no HIP in the translation unit), writes the scenes it
reads and runs it: the reference's per-keyframe local-BA call without Python in the timed region.
    python tools/lba_call_latency_cpp.py [n_scenes] [rounds]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snake_slam_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = os.path.join(ROOT, "snake_slam_amd", "lib")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "lba_call_latency")
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", f"-I{ROOT}/include", f"-I{ROOT}/snake_slam_amd/cpp", f"{ROOT}/tools/cpp/lba_call_latency.cpp",
               f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
        subprocess.run(cmd, check=True)
        for k in range(n):
            sc = synth.ba_scene(seed=1000 + k, outlier_frac=float(os.environ.get("LBA_OUTLIER_FRAC", "0.02")))[0]
            for name, dt in (("pose", np.float64), ("img_const", np.uint8), ("pt", np.float64), ("pt_const", np.uint8), ("obs_img", np.int32),
                             ("obs_pt", np.int32), ("obs_uv", np.float64), ("obs_depth", np.float64), ("obs_weight", np.float64), ("K", np.float64)):
                np.ascontiguousarray(sc[name], dt).tofile(os.path.join(d, f"scene{k}_{name}.bin"))
            np.array([float(sc["bf"])], np.float64).tofile(os.path.join(d, f"scene{k}_bf.bin"))
        r = subprocess.run([exe, d, str(n), str(rounds)], capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        sys.stderr.write(r.stderr[-2000:])
        return r.returncode


if __name__ == "__main__":
    sys.exit(main())
