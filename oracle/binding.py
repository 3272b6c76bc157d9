"""ctypes binding of the CPU oracle (oracle/liboracle.so, oracle/liboracle_omp.so, oracle/_ref/libvpp_ref.so).
TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(ref=True):
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir("/root/reference/vpp"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build(ref=False)
    return ctypes.CDLL(path)


def load(omp=False):
    lib = _load("liboracle_omp.so" if omp else "liboracle.so")
    lib.orc_num_threads.restype = ctypes.c_int
    return lib


def load_ref_omp():
    """The reference's own headers built with its benchmark flags (-O3 -fopenmp -DNDEBUG): the CPU baseline of bench.py; None if never built."""
    path = os.path.join(HERE, "_ref", "libvpp_ref_omp.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


def load_ref_video_extruder():
    """The reference's video_extruder, built with -DNDEBUG (see oracle/ref/ref_video_extruder.cpp); None if never built."""
    path = os.path.join(HERE, "_ref", "libvpp_ref_ve.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


def load_ref():
    """The reference's own headers compiled against shims; None when it was never built (no /root/reference)."""
    path = os.path.join(HERE, "_ref", "libvpp_ref.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None
