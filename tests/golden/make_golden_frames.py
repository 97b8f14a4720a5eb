#!/usr/bin/env python3
"""Mint LZ4-FRAME golden vectors with the system liblz4's LZ4F_compressFrame (the C code family the reference links
through lz4-sys): linked and independent blocks, all block sizes, block/content checksums, content size, fast and HC
levels.  Inputs are regenerated deterministically by the tests (`kind`, `n`), only the frames are stored.
Run in the build container:  python tests/golden/make_golden_frames.py  ->  tests/golden/golden_frames.json"""
import base64, ctypes as C, hashlib, json, os, random

HERE = os.path.dirname(os.path.abspath(__file__))


def content(kind, n):
    if kind == "text":
        t = open(os.path.join(HERE, "plaintext.txt"), "rb").read()
        return (t * (n // len(t) + 1))[:n]
    if kind == "mixed":
        rnd = random.Random(n)
        return bytes(rnd.choice(b"abcdefgh ") for _ in range(n // 2)) + bytes(n - n // 2 - 1000) + bytes(rnd.randrange(256) for _ in range(1000))
    raise ValueError(kind)


class FI(C.Structure):
    _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int), ("frameType", C.c_int),
                ("contentSize", C.c_ulonglong), ("dictID", C.c_uint), ("blockChecksumFlag", C.c_int)]


class PR(C.Structure):
    _fields_ = [("frameInfo", FI), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint), ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]


def main():
    L = C.CDLL("liblz4.so.1")
    L.LZ4_versionNumber.restype = C.c_int
    L.LZ4F_compressFrameBound.restype = C.c_size_t; L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    L.LZ4F_compressFrame.restype = C.c_size_t; L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    vecs = []
    for kind, n in (("text", 0), ("text", 1), ("text", 857), ("text", 65536), ("text", 65537), ("text", 300000), ("mixed", 140000)):
        data = content(kind, n)
        for linked in (1, 0):
            for bs in ((4,) if (n < 100000 or kind == "mixed") else (4, 5, 7)):
                for csum, bsum, csize, level in (((1, 0, 0, 4),) if kind == "mixed" else ((1, 0, 0, 4), (0, 1, 1, 0), (1, 1, 0, 0))):
                    pr = PR()
                    pr.frameInfo.blockSizeID = bs; pr.frameInfo.blockMode = 0 if linked else 1
                    pr.frameInfo.contentChecksumFlag = csum; pr.frameInfo.blockChecksumFlag = bsum
                    pr.frameInfo.contentSize = len(data) if csize else 0; pr.compressionLevel = level
                    cap = L.LZ4F_compressFrameBound(len(data), C.byref(pr)); out = C.create_string_buffer(cap)
                    r = L.LZ4F_compressFrame(out, cap, data, len(data), C.byref(pr))
                    assert r < 1 << 60
                    vecs.append(dict(kind=kind, n=n, linked=linked, bs=bs, content_checksum=csum, block_checksum=bsum, content_size=csize,
                                     level=level, sha256=hashlib.sha256(data).hexdigest(), frame=base64.b64encode(out.raw[:r]).decode()))
    with open(os.path.join(HERE, "golden_frames.json"), "w") as f:
        json.dump(dict(made_by="liblz4 %d LZ4F_compressFrame" % L.LZ4_versionNumber(), vectors=vecs), f)
    print(len(vecs), "frames", os.path.getsize(os.path.join(HERE, "golden_frames.json")), "bytes")


if __name__ == "__main__":
    main()
