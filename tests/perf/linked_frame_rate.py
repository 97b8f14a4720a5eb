import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cj
mb = 64
parts = [oracle.synth_v1(65536, i) for i in range(64)]
data = b"".join(parts[i % 64] for i in range(mb * 16))
r, linked = oracle.lz4_frame_compress(data, 4, 1)
for _ in range(4):
    t = time.perf_counter(); out = cj.lz4.decompress(linked); dt = time.perf_counter() - t
    print("linked frame decompress %.2f ms" % (dt * 1e3))
assert bytes(out) == data
