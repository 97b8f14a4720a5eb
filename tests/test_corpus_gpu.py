"""BASELINE configs[0] on the GPU box: whole-file round trips of benchmark-corpus files through the Python API
(reference benchmarks/test_bench.py:63-64,245-266 — round_trip(compress_block, decompress_block, file)), cross-checked with
the CPU oracle in both directions, plus the two 54 MB synthetic inputs of benchmarks/test_bench.py:38-60 at full size
(the random one seeded here; the reference leaves it unseeded).  The files are data fixtures the reference's benchmarks hold
(tests/golden/corpus, sha256 in manifest.json)."""
import bz2
import hashlib
import json
import os

import numpy as np
import pytest

import cramjam_amd as cramjam
import oracle
from conftest import GOLDEN_DIR

CORPUS = os.path.join(GOLDEN_DIR, "corpus")
MANIFEST = json.load(open(os.path.join(CORPUS, "manifest.json")))["files"]


def _load(name):
    raw = bz2.decompress(open(os.path.join(CORPUS, name + ".bz2"), "rb").read())
    assert len(raw) == MANIFEST[name]["bytes"] and hashlib.sha256(raw).hexdigest() == MANIFEST[name]["sha256"]
    return raw


def test_corpus_fixtures_are_intact():                      # CPU: the fixtures decode to what the manifest pins
    assert len(MANIFEST) >= 6
    for name in MANIFEST:
        _load(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_whole_file_round_trip(name):
    raw = _load(name)
    # LZ4 block: GPU compress -> GPU decompress, GPU compress -> oracle decode, oracle compress -> GPU decode
    c = bytes(cramjam.lz4.compress_block(raw))
    assert int.from_bytes(c[:4], "little") == len(raw)                                  # store_size prefix (src/lz4.rs:113-131)
    assert bytes(cramjam.lz4.decompress_block(c)) == raw
    assert oracle.lz4_block_decompress(c, len(raw), True) == (len(raw), raw)
    assert bytes(cramjam.lz4.decompress_block(oracle.lz4_block_compress(raw)[1])) == raw
    out = np.zeros(len(raw), dtype=np.uint8)
    assert cramjam.lz4.decompress_block_into(c, out) == len(raw) and out.tobytes() == raw
    # Snappy raw
    s = bytes(cramjam.snappy.compress_raw(raw))
    assert bytes(cramjam.snappy.decompress_raw(s)) == raw
    assert oracle.snappy_decompress(s) == (len(raw), raw)
    assert bytes(cramjam.snappy.decompress_raw(oracle.snappy_compress(raw)[1])) == raw
    # framed containers (rows f-1)
    assert bytes(cramjam.lz4.decompress(cramjam.lz4.compress(raw))) == raw
    assert bytes(cramjam.snappy.decompress(cramjam.snappy.compress(raw))) == raw


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fifty-four-mb-repeating", "fifty-four-mb-random"])
def test_fifty_four_mb_inputs(kind):
    if kind == "fifty-four-mb-repeating":
        raw = b"oh what a beautiful morning, oh what a beautiful day!!" * 1000000
    else:
        raw = np.random.default_rng(54).integers(0, 255, size=54_000_000, dtype=np.uint8).tobytes()
    c = bytes(cramjam.lz4.compress_block(raw))
    assert bytes(cramjam.lz4.decompress_block(c)) == raw
    assert oracle.lz4_block_decompress(c, len(raw), True) == (len(raw), raw)
    s = bytes(cramjam.snappy.compress_raw(raw))
    assert bytes(cramjam.snappy.decompress_raw(s)) == raw
    assert oracle.snappy_decompress(s) == (len(raw), raw)
