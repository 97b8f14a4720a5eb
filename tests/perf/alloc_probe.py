import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, oracle
import cramjam_amd as cj
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = b"".join(parts[i % 64] for i in range(64 * 16))
comp = bytes(cj.snappy.compress(data))
def t(fn, reps=5):
    fn(); b = 1e9
    for _ in range(reps):
        s = time.perf_counter(); fn(); b = min(b, time.perf_counter() - s)
    return b * 1e3
pre = np.zeros(len(data), np.uint8)
print("decompress_into prefaulted       %.2f ms" % t(lambda: cj.snappy.decompress_into(comp, pre)))
print("decompress_into fresh np.empty   %.2f ms" % t(lambda: cj.snappy.decompress_into(comp, np.empty(len(data), np.uint8))))
print("np.empty + fill only             %.2f ms" % t(lambda: np.empty(len(data), np.uint8).fill(0)))
print("decompress -> Buffer             %.2f ms" % t(lambda: cj.snappy.decompress(comp)))
keep = []
print("decompress -> Buffer (kept alive)%.2f ms" % t(lambda: keep.append(cj.snappy.decompress(comp))))
print("Buffer(bytes 64MiB) construct    %.2f ms" % t(lambda: cj.Buffer(data)))
print("bytes(64MiB) copy                %.2f ms" % t(lambda: bytes(memoryview(data))))
