"""Builds the native pieces in-tree with hipcc / gcc.

    python -m raytracer_amd.build            # product library + CLI driver + oracle
    python -m raytracer_amd.build --force    # ignore timestamps

Outputs (git-ignored, shipped to the GPU box with the tree):
    raytracer_amd/libraytracer_amd.so   HIP kernels + C ABI + host front end (gfx950)
    raytracer_amd/bin/rt_driver         command-line host
    oracle/librt_oracle.so, oracle/rt_oracle   CPU checker (tests and cpu_baseline only)
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libraytracer_amd.so")
DRIVER = os.path.join(PKG, "bin", "rt_driver")

# No FMA contraction, no fast-math: the kernels evaluate the reference's FP64 expressions
# operation for operation.
DEVICE_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
HOST_FLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-value"]


def source_id():
    """Identity of the sources the library is built from: sha256 over csrc/*.{hip,hpp,cpp} and the public header,
    16 hex digits.  Compiled into the library (rt_build_id) so that measurements can name the build they belong to."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".cpp"))]
    files.append(os.path.join(ROOT, "include", "raytracer_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP library cannot be built")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_library(force=False):
    hipcc = _hipcc()
    dev_src = os.path.join(CSRC, "rt_device.hip")
    host_src = os.path.join(CSRC, "host_scene.cpp")
    header = os.path.join(ROOT, "include", "raytracer_amd.h")
    deps = [dev_src, host_src, header, os.path.join(CSRC, "host_scene.hpp"), os.path.abspath(__file__)]
    accel_src = os.path.join(CSRC, "accel_build.cpp")
    accel_hdr = os.path.join(CSRC, "accel_build.hpp")
    deps += [accel_src, accel_hdr]
    dev_obj = os.path.join(CSRC, "rt_device.o")
    host_obj = os.path.join(CSRC, "host_scene.o")
    accel_obj = os.path.join(CSRC, "accel_build.o")
    lc_src = os.path.join(CSRC, "lightcam_build.cpp")
    lc_hdr = os.path.join(CSRC, "lightcam_build.hpp")
    lc_obj = os.path.join(CSRC, "lightcam_build.o")
    deps += [lc_src, lc_hdr]
    if force or _newer(lc_obj, [lc_src, lc_hdr, os.path.abspath(__file__)]):
        _run(["g++"] + HOST_FLAGS + ["-c", lc_src, "-o", lc_obj])
    dev_hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    sid = source_id()
    id_file = os.path.join(CSRC, ".build_id")
    built_id = open(id_file).read().strip() if os.path.exists(id_file) else ""
    if force or built_id != sid or _newer(dev_obj, [dev_src, header, os.path.abspath(__file__)] + dev_hdrs):
        _run([hipcc] + DEVICE_FLAGS + ['-DRT_BUILD_ID="%s"' % sid, "-c", dev_src, "-o", dev_obj])
        with open(id_file, "w") as f:
            f.write(sid + "\n")
    group_src = os.path.join(CSRC, "rt_group.hip")
    group_obj = os.path.join(CSRC, "rt_group.o")
    if force or _newer(group_obj, [group_src, header, os.path.abspath(__file__)]):
        _run([hipcc] + DEVICE_FLAGS + ["-c", group_src, "-o", group_obj])
    if force or _newer(host_obj, [host_src, header, os.path.join(CSRC, "host_scene.hpp"), os.path.abspath(__file__)]):
        _run(["g++"] + HOST_FLAGS + ["-c", host_src, "-o", host_obj])
    if force or _newer(accel_obj, [accel_src, accel_hdr, header, os.path.abspath(__file__)]):
        _run(["g++"] + HOST_FLAGS + ["-c", accel_src, "-o", accel_obj])
    if force or _newer(LIB, deps + [dev_obj, host_obj, accel_obj, lc_obj, group_obj]):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, dev_obj, host_obj, accel_obj, lc_obj, group_obj, "-lpthread"])
    return LIB


def build_driver(force=False):
    src = os.path.join(CSRC, "rt_driver.cpp")
    os.makedirs(os.path.dirname(DRIVER), exist_ok=True)
    if force or _newer(DRIVER, [src, LIB]):
        _run(["g++", "-O2", "-std=c++17", src, "-o", DRIVER, "-L" + PKG, "-lraytracer_amd",
              "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + "/opt/rocm/lib"])
    return DRIVER


EXAMPLE = os.path.join(PKG, "bin", "pointer_octree_host")


def build_example(force=False):
    """The compile-tested form of the reference-side integration (examples/pointer_octree_host.cpp)."""
    src = os.path.join(ROOT, "examples", "pointer_octree_host.cpp")
    os.makedirs(os.path.dirname(EXAMPLE), exist_ok=True)
    if force or _newer(EXAMPLE, [src, LIB, os.path.join(ROOT, "include", "raytracer_amd.h")]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", EXAMPLE, "-L" + PKG,
              "-lraytracer_amd", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + "/opt/rocm/lib"])
    return EXAMPLE


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    if force:
        _run(["make", "-C", odir, "clean"])
    _run(["make", "-C", odir])
    return os.path.join(odir, "librt_oracle.so")


def build_all(force=False):
    build_library(force)
    build_driver(force)
    build_example(force)
    build_oracle(force)


if __name__ == "__main__":
    if "--source-id" in sys.argv:
        print(source_id())
    else:
        build_all(force="--force" in sys.argv)
