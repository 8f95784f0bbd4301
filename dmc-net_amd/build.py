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


MANIFEST = os.path.join(PKG, "libdmcnet_hip.manifest.json")


def _sha(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def source_hashes():
    """sha256 of everything the library is built from (every file of csrc/ that is not an object, the public header, this recipe)."""
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith((".o", ".hipfb")))
    deps += [os.path.join(ROOT, "include", "dmcnet_hip.h"), os.path.abspath(__file__)]
    return {os.path.relpath(d, ROOT): _sha(d) for d in deps}


def read_manifest():
    import json
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _stale():
    """True unless the library on disk is the one the manifest describes AND the manifest's source hashes are the tree's.
    (Content, not modification times: a prebuilt library that travelled with a snapshot is recognised -- or refused -- by what
    it was built from.)"""
    if not os.path.exists(LIB):
        return True
    m = read_manifest()
    return m is None or m.get("library_sha256") != _sha(LIB) or m.get("sources") != source_hashes()


def _write_manifest(compiled):
    import json
    import time
    with open(MANIFEST, "w") as f:
        json.dump({"library": os.path.relpath(LIB, ROOT), "library_sha256": _sha(LIB), "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                   "compiled_objects": compiled, "hipcc": HIPCC, "flags": FLAGS, "sources": source_hashes()}, f, indent=1, sort_keys=True)


#: what the last build_library() call of this process did: "up to date" or the list of objects it compiled (recorded for the driver's log)
LAST_BUILD = {"action": None}


MEASURE_LIB = os.path.join(PKG, "libdmcnet_hip_measure.so")


def build_library(force=False, verbose=False, measure=False):
    """Compile every HIP source into dmc-net_amd/libdmcnet_hip.so; returns the path.

    measure=True: the -DDMC_MEASURE build (libdmcnet_hip_measure.so) -- the only one in which the options "gen_ablate",
    "conv_ablate" and "gen_stagger" (parts of a kernel switched off, results wrong) exist.  tools/ load it through
    DMC_HIP_LIB; the package never does."""
    if measure:
        return _build_measure(verbose)
    if not force and not _stale():
        LAST_BUILD["action"] = "up to date (sources and library match the manifest's hashes)"
        return LIB
    objs, compiled = [], []
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
        compiled.append(src)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _write_manifest(compiled)
    LAST_BUILD["action"] = "compiled %d of %d sources and linked: %s" % (len(compiled), len(objs), ", ".join(compiled) or "(objects were current)")
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
