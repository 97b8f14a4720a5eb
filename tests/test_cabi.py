"""The drop-in boundary: libcramjam_hip.so loads and exports every symbol include/cramjam_hip.h declares.
No compute calls here (no GPU needed)."""
import ctypes as C
import os
import re

from conftest import ROOT
from cramjam_amd import _native as N


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cj_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    abi = _declared_symbols("cramjam_hip.h")                  # the drop-in ABI: what a binding of the reference's call sites uses
    dbg = _declared_symbols("cramjam_hip_debug.h")            # test / benchmark utilities and debug counters, declared apart
    assert len(abi) >= 24 and not set(abi) & set(dbg)
    assert all(n.startswith(("cj_bench_", "cj_debug_")) for n in dbg) and not any(n.startswith(("cj_bench_", "cj_debug_")) for n in abi)
    L = C.CDLL(N.LIB_PATH)
    for n in abi + dbg:
        assert hasattr(L, n), "libcramjam_hip.so does not export %s" % n
    # each list is bound as what it is: nothing else is bound, nothing else exported
    assert sorted(N.SYMBOLS) == abi, sorted(set(N.SYMBOLS) ^ set(abi))
    assert sorted(N.BENCH_SYMBOLS) == dbg, sorted(set(N.BENCH_SYMBOLS) ^ set(dbg))
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True).stdout
    c_syms = sorted(ln.split()[-1] for ln in exported.splitlines() if ln.split() and not ln.split()[-1].startswith("_Z") and ln.split()[1] in "TtWw")
    assert c_syms == sorted(abi + dbg), "exported C symbols that no header declares: %s" % sorted(set(c_syms) - set(abi + dbg))


def test_shipped_library_reads_no_pipeline_environment():
    # the knobs of tuning builds (-DCJ_DEBUG_KNOBS) must not be in the product: no getenv of a CJ_ name in the shipped binary
    blob = open(N.LIB_PATH, "rb").read()
    for knob in (b"CJ_FUSED", b"CJ_SLICE_CHUNKS", b"CJ_DEVICE", b"CJ_LZ4F_ONE_WORKGROUP", b"CJ_LZ4F_CHAIN_ONLY", b"CJ_SLAB_PROFILE"):
        assert knob + b"\0" not in blob, knob


def test_pure_helpers_and_error_strings():
    L = N.lib()
    assert L.cj_abi_version() == 1
    assert L.cj_lz4_block_compress_bound(65536, 0) == 65809 and L.cj_lz4_block_compress_bound(65536, 1) == 65813
    assert L.cj_lz4_block_compress_bound(0x7E000001, 0) == 0
    assert L.cj_snappy_raw_max_compress_len(65536) == 76490
    pre = b"\x39\x05\x00\x00rest"
    assert L.cj_lz4_block_prefixed_len(pre, len(pre)) == 1337
    assert L.cj_lz4_block_prefixed_len(b"ab", 2) == -3
    assert L.cj_snappy_raw_decompress_len(b"\xd9\x06", 2) == 857
    assert N.strerror(-7) == "Decompression failed. Input invalid or too long?"
    assert N.strerror(-2) == "Compression failed"
    assert "no CPU fallback" in N.strerror(-100)


def test_no_silent_cpu_path_without_device():
    import torch
    if torch.cuda.is_available():
        return
    L = N.lib()
    out = C.create_string_buffer(64)
    assert L.cj_device_count() == 0
    assert L.cj_lz4_block_compress(b"abc", 3, C.cast(out, C.c_void_p), 64, -1, -1, -1) == -100
    assert L.cj_snappy_raw_decompress(b"\x00", 1, C.cast(out, C.c_void_p), 64) == -100


def test_product_does_not_touch_the_oracle():
    # nothing under cramjam_amd/ may import, link or dlopen oracle/
    pkg = os.path.join(ROOT, "cramjam_amd")
    for dp, _, files in os.walk(pkg):
        if os.path.basename(dp) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "cj_oracle" not in src and "import oracle" not in src and "libcj_oracle" not in src, os.path.join(dp, f)
