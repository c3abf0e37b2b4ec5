"""Builds libkzg_mi355x.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch, no JIT cache.

    python -m zkp_subnet_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libkzg_mi355x.so")
SOURCES = ["msm_sort.hip", "msm_accumulate.hip", "msm_tree.hip", "g1_kernels.hip", "srs_kernels.hip", "fr_ntt.hip", "fr_poly.hip", "lanes.hip", "srs.hip", "pipeline.hip", "serve.hip", "comm.hip", "abi_test.hip", "calibrate.hip",
           "pairing_host.cpp", "finish_host.cpp", "rccl_dl.cpp", "multi_host.cpp", "wire_host.cpp"]
# dev-only prototypes (scripts/proto/), linked only when KZG_WITH_PROTO=1: never part of the shipped library
PROTO_SOURCES = ["../../scripts/proto/baff_proto.hip"]
HEADERS = ["bigint.hip.h", "field.hip.h", "fp28.hip.h", "fr29.hip.h", "g1.hip.h", "msm.hip.h", "msm_dev.hip.h", "fr_kernels.hip.h", "fp_lp.hip.h", "fp_host.h", "rccl_dl.h", "lanebook.h", "ctx.hip.h",
           "../../include/kzg_mi355x.h", "../../include/kzg_mi355x_test.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X prover cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_wire(force: bool = False) -> str:
    """The CPython text codec of the synapse (csrc/wire_py.c): plain C, built with gcc against this interpreter."""
    import sysconfig

    src = os.path.join(CSRC, "wire_py.c")
    out = os.path.join(HERE, "_wire" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if force or _stale(out, [src]):
        # KZG_WIRE_CC / KZG_WIRE_CFLAGS: the sanitizer builds of scripts/sanitize_cpu.sh (clang + -fsanitize=...)
        cc = os.environ.get("KZG_WIRE_CC") or shutil.which("gcc") or shutil.which("cc")
        if not cc:
            raise RuntimeError("gcc not found: cannot build the wire codec extension")
        cflags = os.environ.get("KZG_WIRE_CFLAGS", "-O3").split()
        res = subprocess.run([cc, *cflags, "-shared", "-fPIC", "-pthread", "-I" + sysconfig.get_paths()["include"], src,
                              "-o", out], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("wire codec build failed:\n" + res.stderr[-4000:])
    return out


def build(force: bool = False, extra_flags=()) -> str:
    """The shipped library.  Objects are rebuilt when a source, a header OR the flag set changes (the flags of the last
    build are kept in build/flags.stamp).  KZG_WITH_PROTO=1 builds the dev prototypes into their OWN directory and
    library (build_proto/, libkzg_mi355x_proto.so; load it with KZG_MI355X_LIB=...): it never touches the shipped one;
    KZG_BUILD_TAG=<tag> likewise builds the tree with KZG_EXTRA_HIPCC_FLAGS into build_<tag>/ and ab/<tag>.so."""
    build_wire(force)
    hipcc = _hipcc()
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("KZG_EXTRA_HIPCC_FLAGS", "").split())
    sources = list(SOURCES)
    obj_dir, lib = OBJ, LIB
    tag = os.environ.get("KZG_BUILD_TAG")
    if tag:   # an A/B or timing build (KZG_EXTRA_HIPCC_FLAGS=-D...): its own objects and zkp_subnet_amd/ab/<tag>.so, loaded
        obj_dir, lib = os.path.join(HERE, "build_" + tag), os.path.join(HERE, "ab", tag + ".so")   # with KZG_MI355X_LIB=...
        os.makedirs(os.path.dirname(lib), exist_ok=True)
    if os.environ.get("KZG_WITH_PROTO") == "1":
        sources += PROTO_SOURCES
        extra_flags += ("-DKZG_WITH_PROTO",)
        obj_dir, lib = os.path.join(HERE, "build_proto"), os.path.join(HERE, "libkzg_mi355x_proto.so")
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(obj_dir, "flags.stamp")
    flag_key = " ".join([*FLAGS, *extra_flags])
    try:
        with open(stamp) as f:
            if f.read() != flag_key:
                force = True
    except OSError:
        force = force or any(f.endswith(".o") for f in os.listdir(obj_dir))   # objects of unknown flags
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in sources:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for res in ex.map(lambda cmd: subprocess.run(cmd, capture_output=True, text=True), jobs):
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed:\n" + " ".join(res.args) + "\n" + res.stderr[-4000:])
    with open(stamp, "w") as f:
        f.write(flag_key)
    objs = [os.path.join(obj_dir, os.path.splitext(os.path.basename(s))[0] + ".o") for s in sources]
    if force or jobs or _stale(lib, objs):
        link_flags = [f for f in extra_flags if f.startswith("-fsanitize") or f in ("-shared-libsan", "-fno-gpu-sanitize")]
        res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *link_flags, "-o", lib, *objs],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n" + res.stderr[-4000:])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
