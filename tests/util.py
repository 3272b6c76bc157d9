"""Test-side alias of vpp_amd.synth (seeded synthetic inputs) — kept so that test modules read `from util import ...`."""
from vpp_amd.synth import *  # noqa: F401,F403
from vpp_amd.synth import P, HostImage, DeviceImage, U8, I32, F32  # noqa: F401
