"""Synthetic 2D-3D match sets for the pose-refinement tests (oracle and GPU)."""
import numpy as np

from snake_slam_amd.synth import POSE_CAM as CAM  # noqa: F401
from snake_slam_amd.synth import perturb_pose as perturb  # noqa: F401
from snake_slam_amd.synth import pose_problem as make_problem  # noqa: F401
from snake_slam_amd.synth import quat_to_R, random_pose  # noqa: F401


def pose_error(a, b):
    """(rotation angle [rad], translation distance) between two world->camera poses."""
    Ra, Rb = quat_to_R(a[:4]), quat_to_R(b[:4])
    c = (np.trace(Ra @ Rb.T) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1))), float(np.linalg.norm(a[4:] - b[4:]))
