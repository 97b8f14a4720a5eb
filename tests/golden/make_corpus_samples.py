"""Mint the 64 KiB-chunk SAMPLES of the eight large files of the reference's benchmark corpus (/root/reference/benchmarks/data/
{dickens,mr,nci,ooffice,osdb,reymont,x-ray,xml}.bz2; benchmarks/test_bench.py:38-64 round-trips the whole files).  The twelve small
files travel whole (tests/golden/corpus/*.bz2); of every large one, 12 full 65 536-byte chunks spread evenly over the file are kept —
data, bz2-compressed, sha256-pinned in manifest.json — so that `bench.py --data corpus64k` and tests/test_corpus_gpu.py span all 20
files of SURVEY.md §8(d)'s corpus-64k without carrying 70 MB.  Run here (needs /root/reference); the outputs are committed."""
import bz2
import hashlib
import json
import os

SRC = "/root/reference/benchmarks/data"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus")
LARGE = ["dickens", "mr", "nci", "ooffice", "osdb", "reymont", "x-ray", "xml"]
S, PER = 65536, 12

man_path = os.path.join(DST, "manifest.json")
man = json.load(open(man_path))
man["samples"] = {}
for name in LARGE:
    raw = bz2.decompress(open(os.path.join(SRC, name + ".bz2"), "rb").read())
    n_full = len(raw) // S
    picks = [int(i * (n_full - 1) / (PER - 1)) for i in range(PER)]
    blob = b"".join(raw[p * S:(p + 1) * S] for p in picks)
    open(os.path.join(DST, name + ".sample64k.bz2"), "wb").write(bz2.compress(blob, 9))
    man["samples"][name] = {"file_bytes": len(raw), "file_sha256": hashlib.sha256(raw).hexdigest(), "full_chunks_in_file": n_full,
                            "chunk_bytes": S, "picked_chunks": picks, "bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest()}
    print(name, len(raw), n_full, os.path.getsize(os.path.join(DST, name + ".sample64k.bz2")))
man["samples_note"] = ("<name>.sample64k.bz2 = the listed full 65 536-byte chunks (chunk index in the file) of a large corpus file, concatenated; "
                       "made by tests/golden/make_corpus_samples.py from the reference's benchmarks/data")
json.dump(man, open(man_path, "w"), indent=1)
