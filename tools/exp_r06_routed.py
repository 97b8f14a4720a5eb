"""which chunks of the corpus does the one-kernel path hand to the wavefront kernel?  (phase-profile counter [5] = chunks the workgroup decoder finished)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle
import test_enc2_model as T
from cramjam_amd import _native as N
e = N.Engine(0); L = N.lib()
ph = (C.c_ulonglong * 16)()
for codec, comp in ((N.CODEC_LZ4_BLOCK, lambda c: oracle.lz4_compress_raw(c)[1]), (N.CODEC_SNAPPY_RAW, lambda c: oracle.snappy_compress(c)[1])):
    for name, chunks in T.corpus_files():
        chunks = [c for c in chunks if len(c) == 65536][:12]
        if not chunks: continue
        blobs = [comp(c) for c in chunks]
        L.cj_debug_lds_phase_cycles(ph, 1)
        res, outs = e.batch_host(codec, N.OP_DECOMPRESS, N.FLAG_FORCE_LDS_PER_CHUNK | 0x1000, blobs, [65536] * len(blobs))
        L.cj_debug_lds_phase_cycles(ph, 1)
        ok = all(bytes(o) == c for o, c in zip(outs, chunks))
        print("%s %-28s chunks %2d  through the workgroup decoder %2d  %s" % ("lz4   " if codec == N.CODEC_LZ4_BLOCK else "snappy", name, len(chunks), int(ph[5]), "" if ok else "WRONG BYTES"))
