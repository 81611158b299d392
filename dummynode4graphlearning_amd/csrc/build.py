"""Build libdn_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

usage: python -m dummynode4graphlearning_amd.csrc.build [--force]
hipcc cross-compiles without a GPU; the .so travels to the GPU box with the snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = ["dn_runtime.hip", "dn_segment.hip", "dn_index.hip", "dn_index_local.hip", "dn_conv_index.hip", "dn_graphsum.hip", "dn_rel.hip", "dn_rel_ring.hip", "dn_close.hip", "dn_conv_graph.hip", "dn_chain2.hip", "dn_weights.hip", "dn_rel_f32.hip", "dn_gemm.hip", "dn_norm.hip"]
HEADERS = ["dn_common.h", "dn_internal.h", os.path.join(ROOT, "include", "dn_hip.h")]
LIB = os.path.join(PKG, "libdn_hip.so")
OBJDIR = os.path.join(HERE, "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wno-unused-result",
         "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True, tuning=False):
    """tuning=True (or DN_BUILD_TUNING=1): compile with -DDN_TUNING_ENV so the DN_* experiment knobs of dn_rel.hip are read
    from the environment (tools/ab.sh); the default build has no environment access at all."""
    global FLAGS
    if (tuning or os.environ.get("DN_BUILD_TUNING") == "1") and "-DDN_TUNING_ENV" not in FLAGS:
        FLAGS = FLAGS + ["-DDN_TUNING_ENV"]
    for extra in os.environ.get("DN_BUILD_EXTRA", "").split():         # e.g. -DDN_RING_STATS (diagnostic builds only)
        if extra not in FLAGS:
            FLAGS = FLAGS + [extra]
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(OBJDIR, "stamp.txt")
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s) + ".o")
        d = _digest([s] + hdrs)
        dfile = o + ".sha"
        objs.append(o)
        if force or not os.path.exists(o) or not os.path.exists(dfile) or open(dfile).read() != d:
            jobs.append((s, o, dfile, d))

    def compile_one(job):
        s, o, dfile, d = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print("[dn build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(dfile, "w") as f:
            f.write(d)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB) or force:
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-o", LIB] + objs
        if verbose:
            print("[dn build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write("ok\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
    print(LIB)
