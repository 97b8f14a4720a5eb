"""The encoders' kernels (cramjam_amd/csrc/cj_enc2.hpp, lz4_encode.hip, snappy_encode.hip) against their scalar model
(tests/hostsim/enc2_model.c): byte-identical streams on every input shape, in a small batch and in a large one (thousands of
workgroups in flight, every copy of a chunk must come out the same).  The model's own
streams are checked against the oracle's decoders in tests/test_enc2_model.py; here the kernels' bytes are decoded once more."""
import os

import pytest

import oracle
from enc2_cases import cases, synth
from test_enc2_model import corpus_files, model_lib, model_lz4, model_snappy

pytestmark = pytest.mark.gpu

from cramjam_amd import _native as N  # noqa: E402

LZ4, SNAPPY, ENC = N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW, N.OP_COMPRESS


def caps_for(codec, raws):
    L = N.lib()
    return [L.cj_lz4_block_compress_bound(len(r), 0) if codec == LZ4 else L.cj_snappy_raw_max_compress_len(len(r)) for r in raws]


R = int(os.environ.get("CJ_TEST_ENC2_R", "512"))          # positions per round of the library under test: 256 per wavefront of a chunk, two wavefronts (an experiment build with one: 256)


def expected(codec, raws):
    M = model_lib()
    return [model_lz4(M, r, R) if codec == LZ4 else model_snappy(M, r, R) for r in raws]


def decode(codec, blk, n):
    return oracle.lz4_decompress_raw(blk, n) if codec == LZ4 else oracle.snappy_decompress(blk)


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_small_batch_emits_the_models_bytes(codec):
    e = N.Engine(0)
    cs = cases()
    raws = [r for _, r in cs]
    want = expected(codec, raws)
    res, outs = e.batch_host(codec, ENC, 0, raws, caps_for(codec, raws))
    for (name, raw), r, o, w in zip(cs, res, outs, want):
        assert r == len(w) and bytes(o) == w, (name, r, len(w), next((i for i in range(min(len(o), len(w))) if o[i] != w[i]), -1))
        dr, d = decode(codec, bytes(o), len(raw))
        assert dr == len(raw) and d == raw, name
    e.close()


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_every_copy_in_a_large_batch_emits_the_models_bytes(codec):
    e = N.Engine(0)
    uniq = [r for _, r in cases() if len(r) <= 70000] + [synth(200 + i) for i in range(24)]
    U, n = len(uniq), 6000
    want = expected(codec, uniq)
    raws = [uniq[i % U] for i in range(n)]
    res, outs = e.batch_host(codec, ENC, 0, raws, caps_for(codec, raws))
    bad = [i for i in range(n) if res[i] != len(want[i % U]) or bytes(outs[i]) != want[i % U]]
    assert not bad, (len(bad), bad[:8], [i % U for i in bad[:8]])
    e.close()


@pytest.mark.parametrize("codec", [LZ4, SNAPPY])
def test_every_corpus_chunk_emits_the_models_bytes(codec):
    """real data (the reference's benchmark corpus, every 64 KiB chunk that travels with the tests): the kernels' bytes are the model's, so the
    per-file ratio bounds of tests/test_enc2_model.py hold for the GPU encoders"""
    e = N.Engine(0)
    raws = [c for _, chunks in corpus_files() for c in chunks]
    want = expected(codec, raws)
    res, outs = e.batch_host(codec, ENC, 0, raws, caps_for(codec, raws))
    bad = [i for i in range(len(raws)) if res[i] != len(want[i]) or bytes(outs[i]) != want[i]]
    assert not bad, (len(bad), bad[:8])
    e.close()
