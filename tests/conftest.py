import base64
import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The native artefacts are git-ignored build products.  When they are MISSING (fresh checkout), build them once so
    # that the C-ABI / Buffer tests can load them; an existing build is never touched here (hipcc cross-compiles gfx950
    # without a GPU, ~1 min).
    from cramjam_amd import _build
    if not (os.path.exists(_build.LIB) and os.path.exists(_build.pymod_path())):
        _build.build()
        for name in [k for k in sys.modules if k == "cramjam_amd" or k.startswith("cramjam_amd.")]:
            del sys.modules[name]          # the package was imported before its native module existed: import it afresh
    import oracle
    oracle.build()


def b64d(s):
    return base64.b64decode(s)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_raw(golden):
    """name -> uncompressed bytes.  Vectors stored without `raw` are recovered by decoding the
    stand-in-produced LZ4 stream with the oracle and checking the recorded sha256."""
    import oracle
    out = {}
    for v in golden["vectors"]:
        if "raw" in v:
            raw = b64d(v["raw"])
        else:
            r, raw = oracle.lz4_decompress_raw(b64d(v["lz4"]), v["n"])
            assert r == v["n"]
        assert sha(raw) == v["sha256"], v["name"]
        out[v["name"]] = raw
    return out


@pytest.fixture(scope="session")
def plaintext():
    with open(os.path.join(GOLDEN_DIR, "plaintext.txt"), "rb") as f:
        return f.read()
