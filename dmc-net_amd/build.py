"""Build recipe for libdmcnet_hip.so (gfx950 only, hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdmcnet_hip.so")
#: gen_fused_bwd.hip (the one-launch generator data gradient: measured slower) is part of the --measure build only
MEASURE_ONLY = ["gen_fused_bwd.hip"]
SOURCES = ["gen_tiny.hip", "gen_x3.hip", "gen_fused.hip", "gen_wgrad.hip", "losses.hip", "disc_tail.hip", "bn_act.hip", "prepare.hip", "stem.hip", "conv_nhwc.hip", "conv_small.hip", "conv_x3s.hip", "conv_x3q.hip", "disc_first.hip", "conv3d_bf16.hip", "pool3d_bf16.hip", "bn3d_bf16.hip", "stem3d_bf16.hip", "unit3d.hip", "coviar_post.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(ROOT, "include", "dmcnet_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


MEASURE_LIB = os.path.join(PKG, "libdmcnet_hip_measure.so")


def build_library(force=False, verbose=False, measure=False):
    """Compile every HIP source into dmc-net_amd/libdmcnet_hip.so; returns the path.

    measure=True: the -DDMC_MEASURE build (libdmcnet_hip_measure.so) -- the only one in which the options "gen_ablate",
    "conv_ablate" and "gen_stagger" (parts of a kernel switched off, results wrong) exist.  tools/ load it through
    DMC_HIP_LIB; the package never does."""
    if measure:
        return _build_measure(verbose)
    if not force and not _stale():
        return LIB
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        hdrs = [os.path.join(CSRC, "dmc_common.h"), os.path.join(CSRC, "x3s_common.h"), os.path.join(CSRC, "conv_small.h"), os.path.join(CSRC, "gen_x3.h"), os.path.join(CSRC, "gen_fused.h"), os.path.join(CSRC, "gen_fused_inl.h"), os.path.join(CSRC, "gen_wgrad.h"), os.path.join(ROOT, "include", "dmcnet_hip.h"),
                os.path.abspath(__file__)]
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [path] + hdrs):
            continue                              # object newer than its source and the shared headers
        cmd = [HIPCC] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def _build_measure(verbose):
    objs = []
    for src in SOURCES + MEASURE_ONLY:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, "measure_" + src.replace(".hip", ".o"))
        objs.append(obj)
        deps = [path, os.path.abspath(__file__)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
        if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            continue
        cmd = [HIPCC] + FLAGS + ["-DDMC_MEASURE", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", MEASURE_LIB] + objs)
    return MEASURE_LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, measure="--measure" in sys.argv))
