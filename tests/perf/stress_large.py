"""Stress of the large-stream decoder's inter-workgroup hand-over under uneven load: several host threads decompress large
streams of different shapes through the default engine (one after the other: the engine serialises them) while another
thread keeps a second engine busy with batches on its own stream.  Every byte is checked.  GPU only."""
import os, sys, threading, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle, cramjam_amd as cj
from cramjam_amd import _native as N
SECONDS = float(os.environ.get("SECONDS", "20"))
rnd = random.Random(5)
parts = [oracle.synth_v1(65536, i) for i in range(16)]
shifted = bytes(777) + b"".join(parts[i % 16] for i in range(160))
text = b"".join(b"%d bottles of beer on the wall, %d bottles of beer\n" % (rnd.randrange(977), rnd.randrange(1013)) for _ in range(120000))
work = [("lz4 shifted", cj.lz4.decompress_block, oracle.lz4_compress_raw(shifted)[1], shifted, True),
        ("lz4 text", cj.lz4.decompress_block, oracle.lz4_compress_raw(text)[1], text, True),
        ("snappy shifted", cj.snappy.decompress_raw, oracle.snappy_compress(shifted)[1], shifted, False),
        ("lz4 frame linked", cj.lz4.decompress, oracle.lz4_frame_compress(shifted, 4, 1)[1], shifted, False)]
stop = time.time() + SECONDS
errors = []; counts = {}
def loop(name, fn, blob, want, olen):
    n = 0
    while time.time() < stop and not errors:
        got = bytes(fn(blob, output_len=len(want))) if olen else bytes(fn(blob))
        if got != want:
            first = next(i for i in range(min(len(got), len(want))) if got[i] != want[i]) if len(got) == len(want) else -1
            errors.append((name, n, len(got), first)); return
        n += 1
    counts[name] = n
def batches():
    e = N.Engine(0); n = 0
    blobs = [oracle.lz4_compress_raw(p)[1] for p in parts] * 64
    while time.time() < stop and not errors:
        res, outs = e.batch_host(N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, 0, blobs, [65536] * len(blobs))
        if any(int(r) != 65536 for r in res) or bytes(outs[5]) != parts[5]: errors.append(("batch", n)); return
        n += 1
        time.sleep(rnd.random() * 0.003)
    counts["batch"] = n; e.close()
ths = [threading.Thread(target=loop, args=w) for w in work] + [threading.Thread(target=batches)]
for t in ths: t.start()
for t in ths: t.join()
print("errors:", errors); print("iterations:", counts)
sys.exit(1 if errors else 0)
