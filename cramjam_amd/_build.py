"""Build libcramjam_hip.so (gfx950 kernels + C-ABI engine) in-tree with hipcc.  No torch involved."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcramjam_hip.so")
SOURCES = ["engine.hip", "lz4_decode.hip", "lz4_decode_lanes.hip", "lz4_decode_lds.hip", "big_chunks.hip", "lz4_encode.hip", "snappy_decode.hip", "snappy_encode.hip", "frame_kernels.hip", "frame.hip", "large.hip", "big_parse.hip", "bench_util.hip"]
HEADERS = ["lds_shared.hpp", "big_chunks.hpp", "cj_common.hpp", "cj_engine.hpp", "cj_match.hpp", "cj_enc2.hpp", "crc32c_lanes.hpp", "lane_stream.hpp", "snappy_records.hpp", "parse_grammar.hpp", "big_parse.hpp", "xxh32_host.hpp", "lz4_lane_walk.hpp", os.path.join("..", "..", "include", "cramjam_hip.h"), os.path.join("..", "..", "include", "cramjam_hip_debug.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] + os.environ.get("CJ_EXTRA_HIPCC_FLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=5) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    build_pymod(force=force or bool(jobs), verbose=verbose)
    return LIB


def pymod_path():
    import sysconfig
    return os.path.join(HERE, "_cramjam" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pymod(force=False, verbose=False):
    """C++ CPython host module (csrc/pymod.cpp) linked against libcramjam_hip.so next to it."""
    import sysconfig
    src = os.path.join(CSRC, "pymod.cpp")
    out = pymod_path()
    if not force and not _newer(out, [src, os.path.join(CSRC, "xxh32_host.hpp"), os.path.join(HERE, "..", "include", "cramjam_hip.h"), LIB]):
        return out
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-missing-field-initializers",
           "-I" + sysconfig.get_paths()["include"], "-I" + CSRC, src, "-o", out,
           "-L" + HERE, "-lcramjam_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    if verbose and r.stderr.strip():
        print(r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
