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
_MAN = json.load(open(os.path.join(CORPUS, "manifest.json")))
MANIFEST = _MAN["files"]
SAMPLES = _MAN["samples"]            # 12 full 64 KiB chunks of each of the eight large files (tests/golden/make_corpus_samples.py)


def _load(name):
    raw = bz2.decompress(open(os.path.join(CORPUS, name + ".bz2"), "rb").read())
    assert len(raw) == MANIFEST[name]["bytes"] and hashlib.sha256(raw).hexdigest() == MANIFEST[name]["sha256"]
    return raw


def test_corpus_fixtures_are_intact():                      # CPU: the fixtures decode to what the manifest pins
    assert len(MANIFEST) == 12 and len(SAMPLES) == 8       # all 20 files of /root/reference/benchmarks/data are represented
    for name in MANIFEST:
        _load(name)
    for name, m in SAMPLES.items():
        blob = bz2.decompress(open(os.path.join(CORPUS, name + ".sample64k.bz2"), "rb").read())
        assert len(blob) == m["bytes"] == 65536 * len(m["picked_chunks"]) and hashlib.sha256(blob).hexdigest() == m["sha256"], name


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


_CHUNK_CHECK = r"""
import bz2, json, os, sys
import oracle
from cramjam_amd import _native as N
corpus = sys.argv[1]
PARSE = {"1": N.FLAG_FORCE_FUSED_PARSE, "0": N.FLAG_FORCE_PARSE_KERNEL}[sys.argv[2]]     # where the workgroup decoder's parse stage runs
mf = json.load(open(os.path.join(corpus, "manifest.json")))
man = mf["files"]
chunks = []
for name in sorted(man):
    raw = bz2.decompress(open(os.path.join(corpus, name + ".bz2"), "rb").read())
    chunks += [raw[i:i + 65536] for i in range(0, len(raw), 65536)]          # every chunk, the files' short tails included
for name in sorted(mf["samples"]):                                              # + the sampled chunks of the eight large files
    raw = bz2.decompress(open(os.path.join(corpus, name + ".sample64k.bz2"), "rb").read())
    chunks += [raw[i:i + 65536] for i in range(0, len(raw), 65536)]
assert len(chunks) >= 140
eng = N.Engine(0)
L = N.lib()
for codec, comp, dec in ((N.CODEC_LZ4_BLOCK, lambda c: oracle.lz4_compress_raw(c)[1], lambda b, n: oracle.lz4_decompress_raw(b, n)),
                         (N.CODEC_SNAPPY_RAW, lambda c: oracle.snappy_compress(c)[1], lambda b, n: oracle.snappy_decompress(b))):
    blobs = [comp(c) for c in chunks]
    for flags in (0, N.FLAG_FORCE_LDS_PER_CHUNK, N.FLAG_FORCE_WAVE_PER_CHUNK):
        res, outs = eng.batch_host(codec, N.OP_DECOMPRESS, flags | PARSE, blobs, [len(c) for c in chunks])
        assert [int(r) for r in res] == [len(c) for c in chunks], (codec, flags)
        assert all(bytes(o) == c for o, c in zip(outs, chunks)), (codec, flags)
    caps = [(L.cj_lz4_block_compress_bound(len(c), 0) if codec == N.CODEC_LZ4_BLOCK else L.cj_snappy_raw_max_compress_len(len(c))) for c in chunks]
    res, outs = eng.batch_host(codec, N.OP_COMPRESS, 0, chunks, caps)
    for c, o in zip(chunks, outs):
        assert dec(bytes(o), len(c)) == (len(c), c), codec
print("corpus chunks ok", len(chunks))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["1", "0"], ids=["parse-in-decoder", "parse-kernel+decoder"])
def test_every_corpus_chunk_against_the_oracle(fused):
    """SURVEY.md §8(d) `corpus-64k`: every 64 KiB chunk (and every file's short tail) of the twelve corpus files that travel whole + 12 sampled
    chunks of each of the eight large ones (all 20 files of the reference's benchmarks/data), compressed by the
    oracle's encoders (bit-identical to liblz4 / libsnappy): the GPU decodes each to its input in the default pipeline, with the
    workgroup decoder forced and with the one-wavefront kernel, on both sides of the parse-in-kernel threshold (real text has ~10 000
    sequences per chunk and dependency chains a hundred levels deep); the GPU encoders' blocks decode with the oracle"""
    import subprocess, sys
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", _CHUNK_CHECK, CORPUS, fused], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "corpus chunks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
