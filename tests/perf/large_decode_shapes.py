"""Single-buffer decompress rates of ONE large stream for data of different dependency shape (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cramjam
PIECE = 65536
def run(name, data, codec="lz4"):
    blob = oracle.lz4_compress_raw(data)[1] if codec == "lz4" else oracle.snappy_compress(data)[1]
    fn = (lambda: cramjam.lz4.decompress_block(blob, output_len=len(data))) if codec == "lz4" else (lambda: cramjam.snappy.decompress_raw(blob))
    assert bytes(fn()) == data
    best = 1e9
    for _ in range(2):
        t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t)
    t = time.perf_counter(); (oracle.lz4_decompress_raw(blob, len(data)) if codec == "lz4" else oracle.snappy_decompress(blob, len(data))); cpu = time.perf_counter() - t
    print("%-28s %-6s %4d MiB  ratio %6.2f  GPU %7.3f GB/s (%7.2f ms)   oracle, 1 core %6.3f GB/s" % (
        name, codec, len(data) >> 20, len(data) / len(blob), len(data) / best / 1e9, best * 1e3, len(data) / cpu / 1e9), flush=True)
parts = [oracle.synth_v1(PIECE, i) for i in range(32)]
data = bytes(777) + b"".join(parts[i % 32] for i in range(700))
run("synth-v1, off the slab grid", data); run("synth-v1, off the slab grid", data, "snappy")
text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer\n" % (i % 977, i % 1013) for i in range(600000))
run("text lines (deep chains)", text); run("text lines (deep chains)", text, "snappy")
run("zeros", bytes(40 << 20)); run("zeros", bytes(40 << 20), "snappy")
import json, random
rnd = random.Random(7)
log = b"".join(b"2026-09-28T12:%02d:%02d.%03d INFO worker-%d request id=%08x path=/api/v1/items/%d status=%d latency_ms=%d\n" % (
    rnd.randrange(60), rnd.randrange(60), rnd.randrange(1000), rnd.randrange(16), rnd.getrandbits(32), rnd.randrange(5000),
    rnd.choice([200, 200, 200, 404, 500]), rnd.randrange(900)) for _ in range(300000))
run("log lines", log); run("log lines", log, "snappy")
js = b"".join(json.dumps({"id": i, "name": "user%d" % rnd.randrange(1000), "tags": ["a", "b", rnd.choice("xyz")], "score": rnd.random()}).encode() + b"\n" for i in range(300000))
run("JSON lines", js); run("JSON lines", js, "snappy")
src = b"".join(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cramjam_amd", "csrc", f), "rb").read()
               for f in sorted(os.listdir(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cramjam_amd", "csrc")))
               if f.endswith((".hip", ".hpp", ".cpp"))) * 12
run("C++ source", src); run("C++ source", src, "snappy")
run("random (stored)", random.Random(1).randbytes(32 << 20)); run("random (stored)", random.Random(1).randbytes(32 << 20), "snappy")
