"""round 5: where a round of the encoders' matcher spends its cycles — a -DCJ_ENC_PROFILE build (s_memtime around the phases, summed over
all wavefronts), by resident workgroups per CU.  usage: CJ_HIP_LIB=<variant> CJ_ENC_BLOCKS=<n> python tools/exp_r05_encprofile.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cramjam_amd import _native as N

L = N.lib()
L.cj_debug_enc_profile.restype = C.c_int
L.cj_debug_enc_profile.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
e = N.Engine(0)
S, n = 65536, int(os.environ.get("CHUNKS", "20000"))
raw = e.alloc(n * S)
if os.environ.get("CORPUS_FILE"):          # the chunks of one file of the reference's benchmark corpus, tiled
    os.environ["CJ_CORPUS_FILES"] = os.environ["CORPUS_FILE"]
    import bench
    cc, _ = bench.corpus_chunks(S)
    tile = np.frombuffer(b"".join(cc), np.uint8)
    reps = (n * S + tile.size - 1) // tile.size
    e.h2d(raw, np.tile(tile, reps)[:n * S])
else:
    N.check(L.cj_bench_synth_v1(raw, S, S, 0, n, 0x5EED, None))
bound = L.cj_lz4_block_compress_bound(S, 0)
stride = (bound + 15) & ~15
out = e.alloc(n * stride)
ids = np.arange(n, dtype=np.uint64)
meta = np.concatenate([ids * S, np.full(n, S, np.uint64), ids * stride, np.full(n, stride, np.uint64), np.zeros(n, np.uint64)])
d_meta = e.alloc(meta.nbytes)
e.h2d(d_meta, meta)
buf = (C.c_ulonglong * 16)()
for rep in range(2):
    L.cj_debug_enc_profile(buf, 1)
    e.batch_device(N.CODEC_LZ4_BLOCK, N.OP_COMPRESS, 0, n, raw, d_meta, d_meta + 8 * n, out, d_meta + 16 * n, d_meta + 24 * n, d_meta + 32 * n)
    e.sync()
L.cj_debug_enc_profile(buf, 0)
v = list(buf)
rounds = max(v[0], 1)
names = ["rounds", "probe (verify)", "measure", "select+push", "table reads + inserts", "flush (in select)", "waiting for the other wave", "windows"]
print("%s, blocks/CU %s: rounds per chunk %.1f (wave-rounds), windows per round %.2f" % (os.environ.get("CORPUS_FILE") or "synth-v1", os.environ.get("CJ_ENC_BLOCKS", "default"), v[0] / n, v[7] / rounds))
print("  heads per window %.1f, selection passes per window %.2f, windows that took the serial walk %.2f %%" % (v[10] / max(v[7], 1), v[8] / max(v[7], 1), 100.0 * v[9] / max(v[7], 1)))
print("  cycles per wave-round: " + ", ".join("%s %.0f" % (names[i], v[i] / rounds) for i in (1, 2, 3, 4, 5, 6)) + ", sum %.0f" % (sum(v[i] for i in (1, 2, 3, 4, 6)) / rounds))
