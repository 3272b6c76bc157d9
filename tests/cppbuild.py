"""Build recipes of the C++ test programs under tests/cpp (host contract, device API, single-source lambdas, video_extruder parity).

The programs are compiled HERE — by the CPU tests that check that they build and by __graft_entry__.build() — into tests/cpp/_build, which
travels to the GPU box with the repo like the product's own .so files.  The -m gpu tests then RUN the prebuilt programs: a GPU lease without
g++ / hipcc still runs every parity program.  A program is recompiled only where a compiler exists and it is missing or older than what it is
made from (its source, the drop-in headers, the C header, the product library)."""
import glob
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
OUT = os.path.join(CPP, "_build")
INC = os.path.join(ROOT, "vpp_amd", "include")
CSRC = os.path.join(ROOT, "vpp_amd", "csrc")
REFDIR = os.path.join(ROOT, "oracle", "_ref")
REFLIB_VE = os.path.join(REFDIR, "libvpp_ref_ve.so")


def _deps(src, device):
    d = [src] + glob.glob(os.path.join(INC, "vpp", "**", "*.hh"), recursive=True)
    if device:
        d += [os.path.join(ROOT, "include", "vpp_amd.h"), os.path.join(CSRC, "libvpp_amd.so")]
    return [p for p in d if os.path.exists(p)]


def _fresh(exe, deps):
    return os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(p) for p in deps)


def _run(cmd, exe, deps, compiler):
    if _fresh(exe, deps):
        return exe
    if shutil.which(compiler) is None:
        assert os.path.exists(exe), f"{exe} was not prebuilt and there is no {compiler} on this machine"
        return exe   # prebuilt where the compilers are; run as shipped
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(cmd)
    return exe


def _device_link():
    return ["-L" + CSRC, "-lvpp_amd", "-Wl,-rpath," + CSRC, "-Wl,--allow-shlib-undefined"]


def host_program(src, name, extra=()):
    """g++ host build against the drop-in headers (no device library)."""
    exe, path = os.path.join(OUT, name), os.path.join(CPP, src)
    return _run(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused", "-Wno-unknown-pragmas", "-I" + INC, path, "-o", exe] + list(extra), exe, _deps(path, False), "g++")


def device_program(src, name):
    """g++ build of a host program whose tagged functors / algorithm front-ends call the C ABI (-DVPP_AMD_DEVICE)."""
    exe, path = os.path.join(OUT, name), os.path.join(CPP, src)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused", "-Wno-unknown-pragmas", "-I" + INC, path, "-o", exe, "-DVPP_AMD_DEVICE", "-I" + os.path.join(ROOT, "include"),
           "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")] + _device_link()
    deps = _deps(path, True) + [p for p in [os.path.join(ROOT, "oracle", "liboracle.so")] if os.path.exists(p)]
    if os.path.exists(REFLIB_VE):  # the real reference, where it was built
        cmd += ["-DHAVE_VPP_REF", "-L" + REFDIR, "-lvpp_ref_ve", "-Wl,-rpath," + REFDIR]
        deps.append(REFLIB_VE)
    return _run(cmd, exe, deps, "g++")


def single_source_program(src, name):
    """The user's translation unit compiled by hipcc: -DVPP_AMD_DEVICE -DVPP_AMD_HIPCC turns opaque pixel_wise lambdas into gfx950 kernels."""
    exe, path = os.path.join(OUT, name), os.path.join(CPP, src)
    cmd = ["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-result", "-Wno-unused-command-line-argument", "-DVPP_AMD_DEVICE", "-DVPP_AMD_HIPCC",
           "-I" + INC, "-I" + os.path.join(ROOT, "include"), path, "-o", exe, "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")] + _device_link()
    return _run(cmd, exe, _deps(path, True) + [p for p in [os.path.join(ROOT, "oracle", "liboracle.so")] if os.path.exists(p)], "hipcc")


def video_extruder_parity():
    """tests/cpp/video_extruder_parity.cc against the reference's own video_extruder headers (oracle/_ref/libvpp_ref_ve.so): no other checker, no skip."""
    assert os.path.exists(REFLIB_VE), ("oracle/_ref/libvpp_ref_ve.so is missing: build it where /root/reference exists (make -C oracle ref); "
                                       "the video_extruder parity check has no other checker and does not skip")
    exe, path = os.path.join(OUT, "video_extruder_parity"), os.path.join(CPP, "video_extruder_parity.cc")
    cmd = ["g++", "-std=c++17", "-O2", "-w", "-DVPP_AMD_DEVICE", "-I" + INC, "-I" + os.path.join(ROOT, "include"), path, "-o", exe,
           "-L" + REFDIR, "-lvpp_ref_ve", "-Wl,-rpath," + REFDIR] + _device_link()
    return _run(cmd, exe, _deps(path, True) + [REFLIB_VE], "g++")


def build_all():
    """Everything the -m gpu tests run (called by __graft_entry__.build(), after the product library and oracle/_ref exist)."""
    out = [device_program("device_api_test.cc", "device_api_test"), single_source_program("device_lambda_test.cc", "device_lambda_test")]
    if os.path.exists(REFLIB_VE):
        out.append(video_extruder_parity())
    return out
