"""The per-lane LZ4 walker of the HIP kernels (cramjam_amd/csrc/lz4_lane_walk.hpp: lane decoder, parse kernel logic)
compiled for the HOST with shims and run under AddressSanitizer on exact-size buffers, against the oracle and
the golden vectors.  Catches logic and bounds bugs of the kernel source without a GPU (it did: the first lane
kernel had a u32 wrap in its tail copy).  CPU only; the product never links this."""
import ctypes as C
import os
import random
import subprocess
import sys

import pytest

import oracle
from conftest import ROOT, b64d

SIM_DIR = os.path.join(ROOT, "tests", "hostsim")
SIM_SO = os.path.join(SIM_DIR, "libsim_lane_walk.so")

CHILD = r"""
import ctypes as C, json, random, sys
sys.path.insert(0, %(root)r)
import oracle
from base64 import b64decode as b64d
S = C.CDLL(%(so)r); S.sim_walk.restype = C.c_int64
S.sim_walk.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
g = json.load(open(%(golden)r))
bad = 0
def chk(blob, cap, tag):
    global bad
    out = C.create_string_buffer(max(cap, 1)); ns1 = C.c_uint32(0); ns2 = C.c_uint32(0)
    r = S.sim_walk(blob, len(blob), out, cap, 1, C.byref(ns1))
    rp = S.sim_walk(blob, len(blob), None, cap, 0, C.byref(ns2))            # parse-only walk: same verdict, size, count
    er, eo = oracle.lz4_decompress_raw(blob, cap)
    ok = (r < 0) == (er < 0) and (r < 0 or (r == er and out.raw[:r] == eo)) and rp == r and (r < 0 or ns1.value == ns2.value)
    if not ok:
        bad += 1; print("MISMATCH", tag, cap, r, rp, er)
for v in g["vectors"]:
    blob = b64d(v["lz4"])
    for extra in (0, 5, 16, 31, 32, 77): chk(blob, v["n"] + extra, v["name"])
for m in g["malformed_lz4"]: chk(b64d(m["data"]), m["cap"], (m["src"], m["kind"], m["k"]))
random.seed(3)
for t in range(150):
    n = random.choice([0, 1, 5, 13, 40, 100, 1000, 5000, 70000]); alpha = random.choice([2, 4, 16, 256])
    raw = bytes(random.randrange(alpha) for _ in range(n))
    if random.random() < 0.5 and n > 10: raw = (raw[:random.randrange(1, 20)] * n)[:n]
    _, blob = oracle.lz4_compress_raw(raw)
    for cap in (n, n + random.randrange(1, 40)): chk(blob, cap, ("fuzz", t))
    if len(blob) > 3:
        b = bytearray(blob); i = random.randrange(len(b)); b[i] ^= 1 << random.randrange(8); chk(bytes(b), n + 8, ("fuzzbad", t))
for i in range(6):
    raw = oracle.synth_v1(65536, i); _, blob = oracle.lz4_compress_raw(raw); chk(blob, 65536, ("synth", i)); chk(blob, 65586, ("synth+", i))
print("HOSTSIM bad=%%d" %% bad)
"""


def _build():
    hdr = open(os.path.join(ROOT, "cramjam_amd", "csrc", "lz4_lane_walk.hpp")).read()
    body = hdr.replace('#pragma once\n', '').replace('#include "cj_common.hpp"\n', '')
    open(os.path.join(SIM_DIR, "lz4_lane_walk_body.inc"), "w").write(body)
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC",
                           "-o", SIM_SO, os.path.join(SIM_DIR, "sim_lane_walk.cpp")], cwd=SIM_DIR)


def test_lane_walker_host_simulation_under_asan():
    _build()
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    code = CHILD % dict(root=ROOT, so=SIM_SO, golden=os.path.join(ROOT, "tests", "golden", "golden_vectors.json"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "HOSTSIM bad=0" in r.stdout, r.stdout[-3000:]
