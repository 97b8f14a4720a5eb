"""Where the time goes between the completion flags of consecutive slabs (LZ4 frame with linked blocks = every slab waits for
its predecessor).  Needs a build with -DCJ_SLAB_TRACE:
    CJ_EXTRA_HIPCC_FLAGS=-DCJ_SLAB_TRACE python -c "from cramjam_amd import _build; _build.build(force=True)"
"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, oracle, cramjam_amd as cj
from cramjam_amd import _native as N
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = b"".join(parts[i % 64] for i in range(16 * 16))
r, linked = oracle.lz4_frame_compress(data, 4, 1)
for _ in range(2): out = cj.lz4.decompress(linked)
L = C.CDLL(N.lib()._name)
n = 256
buf = np.zeros(n * 8, dtype=np.uint64)
assert L.cj_debug_slab_trace(C.c_void_p(buf.ctypes.data), n) == 0
t = buf.reshape(n, 8).astype(np.int64)
names = ["flag seen", "acquire done", "wave0 cross done", "D3 done", "stores acked", "flag stored"]
# steps relative to predecessor's flag-stored time (100 MHz clock -> 10 ns units)
for k in range(6):
    d = (t[2:200, k] - t[1:199, 5]) * 10e-3
    print("%-18s after predecessor's flag: median %6.2f us  p90 %6.2f" % (names[k], np.median(d), np.percentile(d, 90)))
print("chain step (flag to flag): median %.2f us" % np.median((t[2:200, 5] - t[1:199, 5]) * 10e-3))
