"""cj_batch_host on a LARGE batch (engine.hip: batch_host_sliced — slices that overlap packing, both directions of the
link, the kernels and the scattering): every result and every byte equals what the same chunks give in batches small enough for the
one-shot path, and the oracle's bytes.  GPU only."""
import random

import numpy as np
import pytest

import oracle
from cramjam_amd import _native as N

pytestmark = pytest.mark.gpu


def _streams(codec):
    rnd = random.Random(5)
    raws = [oracle.synth_v1(65536, i) for i in range(48)]
    raws += [bytes(rnd.getrandbits(8) for _ in range(3000)), b"", b"x" * 70000, oracle.synth_v1(1000, 3), bytes(65536)]
    enc = oracle.lz4_compress_raw if codec == N.CODEC_LZ4_BLOCK else oracle.snappy_compress
    comp = [enc(r)[1] for r in raws]
    return raws, comp


@pytest.mark.parametrize("codec", [N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW])
def test_sliced_host_batch_equals_the_one_shot_path(codec):
    raws, comp = _streams(codec)
    rnd = random.Random(9)
    n = 3400                                          # ~215 MB of output + ~135 MB of input: above the slicing threshold
    pick = [rnd.randrange(len(raws)) for _ in range(n)]
    ins = [comp[k] for k in pick]
    caps = [len(raws[k]) for k in pick]
    for i in rnd.sample(range(n), 40):                # damaged streams, capacities that do not fit, empty inputs
        kind = rnd.randrange(3)
        if kind == 0 and len(ins[i]) > 8:
            b = bytearray(ins[i]); b[rnd.randrange(len(b))] ^= 0x5A; ins[i] = bytes(b)
        elif kind == 1 and caps[i] > 2: caps[i] -= 1 + rnd.randrange(min(caps[i] - 1, 50))
        else: ins[i] = b""
    eng = N.Engine(0)
    res, outs = eng.batch_host(codec, N.OP_DECOMPRESS, 0, ins, caps)
    assert len(res) == n
    # the same chunks in batches below the threshold (one-shot path)
    step = 400
    for a in range(0, n, step):
        r2, o2 = eng.batch_host(codec, N.OP_DECOMPRESS, 0, ins[a:a + step], caps[a:a + step])
        assert res[a:a + step] == r2, "results differ in [%d, %d)" % (a, a + step)
        assert outs[a:a + step] == o2
    good = [i for i in range(n) if ins[i] is comp[pick[i]] and caps[i] == len(raws[pick[i]]) and len(raws[pick[i]]) > 0]
    assert len(good) > n - 200
    for i in good[::97]:
        assert res[i] == len(raws[pick[i]]) and outs[i] == raws[pick[i]]
    eng.close()


@pytest.mark.parametrize("codec", [N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW])
def test_sliced_host_batch_compress(codec):
    raws, _ = _streams(codec)
    rnd = random.Random(11)
    n = 2600
    pick = [rnd.randrange(len(raws)) for _ in range(n)]
    ins = [raws[k] for k in pick]
    L = N.lib()
    caps = [(L.cj_lz4_block_compress_bound(len(r), 0) if codec == N.CODEC_LZ4_BLOCK else L.cj_snappy_raw_max_compress_len(len(r))) for r in ins]
    for i in rnd.sample(range(n), 20): caps[i] = max(1, caps[i] // 3)          # (LZ4: "Compression failed" where the result does not fit; Snappy refuses a short buffer)
    eng = N.Engine(0)
    res, outs = eng.batch_host(codec, N.OP_COMPRESS, 0, ins, caps)
    step = 400
    for a in range(0, n, step):
        r2, o2 = eng.batch_host(codec, N.OP_COMPRESS, 0, ins[a:a + step], caps[a:a + step])
        assert res[a:a + step] == r2
        assert outs[a:a + step] == o2
    dec = (lambda b, u: oracle.lz4_decompress_raw(b, u)) if codec == N.CODEC_LZ4_BLOCK else (lambda b, u: oracle.snappy_decompress(b))
    for i in range(0, n, 131):
        if res[i] > 0: assert dec(outs[i], len(ins[i])) == (len(ins[i]), ins[i])
    eng.close()


@pytest.mark.parametrize("codec", [N.CODEC_LZ4_BLOCK, N.CODEC_SNAPPY_RAW])
def test_batch_into_one_output_buffer(codec):
    """cramjam_amd.batch.*(out=...): the outputs land back to back in ONE caller buffer, results and bytes as with bytes objects"""
    from cramjam_amd import batch
    raws, comp = _streams(codec)
    rnd = random.Random(3)
    pick = [rnd.randrange(len(raws)) for _ in range(700)]
    ins = [comp[k] for k in pick]
    lens = [len(raws[k]) for k in pick]
    ins[5] = ins[5][:-3] if len(ins[5]) > 3 else ins[5]            # a damaged stream in the middle
    if codec == N.CODEC_LZ4_BLOCK:
        res0, outs0 = batch.lz4_decompress_blocks(ins, lens)
        out = bytearray(sum(lens))
        res1, outs1 = batch.lz4_decompress_blocks(ins, lens, out=out)
    else:
        res0, outs0 = batch.snappy_decompress_raw_many(ins)
        out = np.zeros(sum(max(len(o), 1) for o in raws) * 20, np.uint8)      # (more than enough; capacities come from the preambles)
        res1, outs1 = batch.snappy_decompress_raw_many(ins, out=out)
    assert res0 == res1
    assert all(bytes(a) == b for a, b in zip(outs1, outs0))
    assert isinstance(outs1[0], memoryview)
    # compress into one buffer, decode the views again
    chunks = [raws[k] for k in pick[:300]]
    if codec == N.CODEC_LZ4_BLOCK:
        L = N.lib()
        buf = bytearray(sum(L.cj_lz4_block_compress_bound(len(c), 0) for c in chunks))
        r, views = batch.lz4_compress_blocks(chunks, store_size=False, out=buf)
        back = [oracle.lz4_decompress_raw(bytes(v), len(c))[1] for v, c in zip(views, chunks)]
    else:
        L = N.lib()
        buf = bytearray(sum(L.cj_snappy_raw_max_compress_len(len(c)) for c in chunks))
        r, views = batch.snappy_compress_raw_many(chunks, out=buf)
        back = [oracle.snappy_decompress(bytes(v))[1] for v in views]
    assert back == chunks
    with pytest.raises(ValueError):
        batch.lz4_decompress_blocks(ins[:4], lens[:4], out=bytearray(10)) if codec == N.CODEC_LZ4_BLOCK else batch.snappy_compress_raw_many(chunks[:4], out=bytearray(10))
