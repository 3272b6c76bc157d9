"""Pyramid helpers for the tests: the device / keypoint helpers of vpp_amd.pyr plus the same construction on the CPU oracle."""
from vpp_amd.pyr import *  # noqa: F401,F403
from vpp_amd.pyr import level_dims, Keypoint, KP_DTYPE  # noqa: F401
from oracle.pyramid import host_pyramid, host_grad_pyramid  # noqa: F401
