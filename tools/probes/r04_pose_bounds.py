"""Builds the variant libraries tools/probes/r04_pose_bounds.sh times (CPU, cross-compiled)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from snake_slam_amd.build import build_variant

VARIANTS = {
    "pose_dppold": {"SNK_DPP_OLD_INIT": 1},
    "pose_w3": {"SNK_POSE_MIN_WAVES": 3},
    "pose_stub1": {"SNK_POSE_STUB_SOLVE": 1},
    "pose_stub2": {"SNK_POSE_STUB_SOLVE": 2},
}
for name in (sys.argv[1:] or VARIANTS):
    print(build_variant(name, VARIANTS[name], "pose.hip"))
