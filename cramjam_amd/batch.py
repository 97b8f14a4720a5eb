"""Batch extension: many independent chunks per call (what the GPU is for).  Host buffers in, host
buffers out; chunks shard round-robin over the given engines (one per GPU), no collective."""
from concurrent.futures import ThreadPoolExecutor

from . import _native as N

_engines = {}


def _engine(device):
    if device not in _engines:
        _engines[device] = N.Engine(device)
    return _engines[device]


def _run(codec, op, flags, inputs, out_caps, devices):
    devices = list(devices) if devices is not None else [0]
    n = len(inputs)
    if len(devices) == 1:
        return _engine(devices[0]).batch_host(codec, op, flags, inputs, out_caps)
    shards = [list(range(g, n, len(devices))) for g in range(len(devices))]   # chunk i -> gpu i mod G

    def work(g):
        idx = shards[g]
        return _engine(devices[g]).batch_host(codec, op, flags, [inputs[i] for i in idx], [out_caps[i] for i in idx])
    with ThreadPoolExecutor(len(devices)) as ex:
        parts = list(ex.map(work, range(len(devices))))
    res, outs = [None] * n, [None] * n
    for g, (r, o) in enumerate(parts):
        for k, i in enumerate(shards[g]):
            res[i], outs[i] = r[k], o[k]
    return res, outs


def lz4_decompress_blocks(blocks, output_lens, store_size=False, devices=None):
    """decode many LZ4 blocks; returns (results, outputs) with results[i] = length or a negative CJ_E_* code"""
    return _run(N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, N.FLAG_LZ4_SIZE_PREFIX if store_size else 0, blocks, output_lens, devices)


def lz4_compress_blocks(chunks, store_size=True, devices=None):
    L = N.lib()
    caps = [L.cj_lz4_block_compress_bound(len(c), 1 if store_size else 0) for c in chunks]
    return _run(N.CODEC_LZ4_BLOCK, N.OP_COMPRESS, N.FLAG_LZ4_SIZE_PREFIX if store_size else 0, chunks, caps, devices)


def snappy_decompress_raw_many(blocks, devices=None):
    L = N.lib()
    import ctypes as C
    caps = []
    for b in blocks:
        b = bytes(b)
        caps.append(max(L.cj_snappy_raw_decompress_len(C.cast(C.c_char_p(b), C.c_void_p), len(b)), 0))
    return _run(N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS, 0, blocks, caps, devices)


def snappy_compress_raw_many(chunks, devices=None):
    L = N.lib()
    caps = [L.cj_snappy_raw_max_compress_len(len(c)) for c in chunks]
    return _run(N.CODEC_SNAPPY_RAW, N.OP_COMPRESS, 0, chunks, caps, devices)
