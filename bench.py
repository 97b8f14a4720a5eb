#!/usr/bin/env python3
"""bench.py — LZ4-block decompress throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over the whole device-resident batch: N chunks of 64 KiB
(synth-v1) compressed with the reference's C code family (system liblz4
LZ4_compress_default; falls back to this engine's own GPU encoder if the library is absent), already in
HBM when the timed region starts, decoded by ONE batch submission (parse kernel -> workgroup-per-chunk LDS
decoder -> wave kernel on routed chunks, all on the engine's stream) into distinct outputs.

`--gpus N` runs N ranks, one per GPU (the script re-launches itself under torch.distributed.run when it was
started plainly; the driver's own torchrun command works as well).  Weak scaling: rank r owns the chunks with
global index i = k*N + r (round-robin, no data-path collective); with N > 1 every GPU holds 125 000 chunks
(8 x 125 000 = BASELINE configs[3]), with N = 1 100 000 (configs[1]).

Every run verifies its whole output outside the timed region: decoded chunks are compared with the generated data on the
device; `--op compress` / `--op roundtrip` (configs[2]) decode every compressed chunk again and compare it with its input.

`--workload mixed256k` is BASELINE configs[4]: 256 KiB chunks, even chunk indices LZ4-block / odd Snappy-raw,
one engine (= one HIP stream) per codec per GPU running concurrently.

Prints ONE JSON line on rank 0 (see the driver contract in the task description):
  value        = uncompressed GB/s over all ranks (sum of bytes / max wall time over ranks)
  roofline     = algorithmic bytes (compressed read + uncompressed written) per step / mean step duration measured
                 with HIP events on the engine's stream, vs the 8 TB/s HBM peak; `traffic` = HBM bytes per step from
                 two extra rocprofv3 passes of this same command (--pmc FETCH_SIZE / --pmc WRITE_SIZE), or null
  cpu_baseline = CPU decoders on this host's cores over the same unique chunks (rank 0, N=1 only): the oracle's C
                 restatement ("port") and, when the host has it, liblz4's LZ4_decompress_safe ("liblz4"), each with
                 all cores and with one
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=None, help="chunks per GPU (default: 100 000 at --gpus 1 = configs[1]; 125 000 at --gpus N = configs[3] at N=8; mixed256k: 16 384)")
    ap.add_argument("--chunk-bytes", type=int, default=None)
    ap.add_argument("--unique", type=int, default=8192, help="distinct chunks generated per GPU; replicated device-side to distinct addresses")
    ap.add_argument("--codec", default="lz4", choices=["lz4", "snappy"])
    ap.add_argument("--data", default="synth-v1", choices=["synth-v1", "corpus64k"],
                    help="corpus64k: the full 64 KiB chunks of the reference's benchmark files that travel with the tests (tests/golden/corpus), tiled to --chunks (SURVEY.md §8d)")
    ap.add_argument("--op", default="decompress", choices=["decompress", "compress", "roundtrip"],
                    help="roundtrip = one compress + one decompress of the batch per step (configs[2] with --codec snappy)")
    ap.add_argument("--workload", default="default", choices=["default", "mixed256k"])
    ap.add_argument("--in-flight", type=int, default=1, help="batches in flight: step k is submitted to engine k mod N (own stream, outputs, results)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default="auto", choices=["auto", "on", "off"],
                    help="HBM bytes per step from two extra rocprofv3 --pmc passes of this command (auto: only for the default 1-GPU workload)")
    ap.add_argument("--compressor", default="auto", choices=["auto", "liblz4", "gpu"])
    ap.add_argument("--phase-profile", action="store_true", help="debug: per-phase cycle counters of the LDS decoder")
    ap.add_argument("--experiment-no-verify", action="store_true",
                    help="kernel experiments that deliberately produce wrong bytes (a phase switched off): skips the comparison and marks the line invalid")
    ap.add_argument("--parse", default="auto", choices=["auto", "fused", "kernel"], help="debug: the workgroup decoder's parse stage inside the decoder kernel / as its own kernel at any batch size (CJ_FLAG_FORCE_FUSED_PARSE / _PARSE_KERNEL)")
    ap.add_argument("--lz4-mode", default="auto", choices=["auto", "wave", "lane", "lds"],
                    help="LZ4 decoder mapping override (results identical; auto = engine default)")
    return ap.parse_args()


def load_liblz4():
    for name in ("liblz4.so.1", "/lib/x86_64-linux-gnu/liblz4.so.1", "/opt/conda/lib/liblz4.so.1"):
        try:
            L = C.CDLL(name)
            L.LZ4_compress_default.restype = C.c_int
            L.LZ4_compress_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            return L, "liblz4 %d (system LZ4_compress_default)" % L.LZ4_versionNumber()
        except OSError:
            continue
    return None, None


def relaunch(args):
    """`python bench.py --gpus N` started plainly: become N ranks under torch.distributed.run (one per GPU)."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d needs %d visible GPUs, found %d (no silent fallback to fewer)" % (args.gpus, args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


class Batch:
    """one codec's device-resident batch on one engine: inputs, descriptors, outputs"""
    pass


def corpus_chunks(S):
    """the benchmark corpus of the reference (benchmarks/data, benchmarks/test_bench.py:38-64) as S-byte chunks, in file order: every full
    chunk of the twelve files that travel whole under tests/golden/corpus, and the 12 sampled full chunks of each of the eight large
    files (<name>.sample64k.bz2, tests/golden/make_corpus_samples.py) — all 20 files, sha256-pinned by manifest.json"""
    import bz2
    import hashlib
    import json
    d = os.path.join(ROOT, "tests", "golden", "corpus")
    mf = json.load(open(os.path.join(d, "manifest.json")))
    man, samples = mf["files"], mf.get("samples", {})
    out = []
    only = os.environ.get("CJ_CORPUS_FILES")                  # (experiments: a comma-separated subset)
    names = [n for n in sorted(list(man) + list(samples)) if not only or n in only.split(",")]
    for name in names:
        if name in man:
            raw = bz2.decompress(open(os.path.join(d, name + ".bz2"), "rb").read())
            assert hashlib.sha256(raw).hexdigest() == man[name]["sha256"], name
        else:
            assert samples[name]["chunk_bytes"] % S == 0, "the large files are carried as %d-byte chunk samples" % samples[name]["chunk_bytes"]
            raw = bz2.decompress(open(os.path.join(d, name + ".sample64k.bz2"), "rb").read())
            assert hashlib.sha256(raw).hexdigest() == samples[name]["sha256"], name
        out += [raw[i:i + S] for i in range(0, len(raw) - S + 1, S)]
    return out, names


def load_libsnappy():
    """the C++ snappy library through its C API (snappy-c.h): the lineage the reference's `snap` crate ports and is tested against"""
    for name in ("libsnappy.so.1", "/opt/conda/lib/libsnappy.so.1", "/usr/lib/x86_64-linux-gnu/libsnappy.so.1", "libsnappy.so"):
        try:
            L = C.CDLL(name)
            L.snappy_compress.restype = C.c_int
            L.snappy_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
            L.snappy_max_compressed_length.restype = C.c_size_t
            L.snappy_max_compressed_length.argtypes = [C.c_size_t]
            return L, "libsnappy (system snappy_compress, %s)" % os.path.basename(name)
        except (OSError, AttributeError):
            continue
    return None, None


def build_batch(N, L, eng, dev, codec, op_dec, S, U, NCH, first_index, compressor, torch, np, data="synth-v1"):
    """U unique chunks — synth-v1 (indices first_index ..) generated on the device, or the corpus chunks — compressed
    (decompress workloads), replicated device-side to NCH chunks at distinct addresses.  Returns a Batch."""
    b = Batch()
    b.codec, b.dec, b.S, b.U, b.NCH, b.eng = codec, op_dec, S, U, NCH, eng
    if data == "corpus64k":
        chunks, _ = corpus_chunks(S)
        assert len(chunks) >= U
        raw = torch.from_numpy(np.frombuffer(b"".join(chunks[:U]), dtype=np.uint8).copy()).to(dev)
    else:
        raw = torch.empty(U * S, dtype=torch.uint8, device=dev)
        N.check(L.cj_bench_synth_v1(raw.data_ptr(), S, S, first_index, U, 0x5EED, None))
    torch.cuda.synchronize()
    b.raw = raw
    bound = L.cj_lz4_block_compress_bound(S, 0) if codec == N.CODEC_LZ4_BLOCK else L.cj_snappy_raw_max_compress_len(S)
    stride_c = (bound + 15) & ~15
    b.stride_c = stride_c
    b.comp_name = None
    b.raw_h = None
    reps = (NCH + U - 1) // U
    ids = np.arange(NCH, dtype=np.uint64)
    if op_dec:
        b.raw_h = raw.cpu().numpy()
        lz4lib = snlib = None
        if codec == N.CODEC_LZ4_BLOCK and compressor in ("auto", "liblz4"):
            lz4lib, b.comp_name = load_liblz4()
        if codec == N.CODEC_SNAPPY_RAW and compressor in ("auto", "liblz4"):       # (the host library of the codec's lineage)
            snlib, b.comp_name = load_libsnappy()
        if lz4lib is not None or snlib is not None:
            from concurrent.futures import ThreadPoolExecutor
            comp_h = np.zeros(U * stride_c, dtype=np.uint8)
            clen = np.zeros(U, dtype=np.uint64)

            def work(i):
                if lz4lib is not None:
                    r = lz4lib.LZ4_compress_default(b.raw_h.ctypes.data + i * S, comp_h.ctypes.data + i * stride_c, S, stride_c)
                    assert r > 0
                else:
                    n = C.c_size_t(stride_c)
                    assert snlib.snappy_max_compressed_length(S) <= stride_c
                    assert snlib.snappy_compress(b.raw_h.ctypes.data + i * S, S, comp_h.ctypes.data + i * stride_c, C.byref(n)) == 0
                    r = n.value
                clen[i] = r
            with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
                list(ex.map(work, range(U)))
        else:
            b.comp_name = "cramjam_amd GPU encoder (%s)" % ("lz4" if codec == N.CODEC_LZ4_BLOCK else "snappy")
            comp_d = torch.zeros(U * stride_c, dtype=torch.uint8, device=dev)
            meta = torch.tensor(np.concatenate([np.arange(U, dtype=np.uint64) * S, np.full(U, S, np.uint64),
                                                np.arange(U, dtype=np.uint64) * stride_c, np.full(U, stride_c, np.uint64),
                                                np.zeros(U, np.uint64)]).view(np.int64), device=dev)
            p = meta.data_ptr()
            torch.cuda.synchronize()
            eng.batch_device(codec, N.OP_COMPRESS, 0, U, raw.data_ptr(), p, p + 8 * U, comp_d.data_ptr(), p + 16 * U, p + 24 * U, p + 32 * U)
            eng.sync()
            clen = meta[4 * U:].cpu().numpy().view(np.int64).astype(np.uint64)
            assert (clen > 0).all()
            comp_h = comp_d.cpu().numpy()
            del comp_d
        # pack the unique compressed chunks tightly (16 B aligned) on the host, upload, replicate on the device
        uoff = np.zeros(U, dtype=np.uint64)
        pos = 0
        for i in range(U):
            uoff[i] = pos
            pos += (int(clen[i]) + 15) & ~15
        packed_total = pos
        packed_h = np.zeros(packed_total, dtype=np.uint8)
        for i in range(U):
            n = int(clen[i])
            packed_h[int(uoff[i]):int(uoff[i]) + n] = comp_h[i * stride_c:i * stride_c + n]
        del comp_h
        b.packed_h, b.uoff, b.clen = packed_h, uoff, clen
        packed = torch.from_numpy(packed_h).to(dev)
        b.cin = packed.repeat(reps)                      # reps distinct copies in HBM
        in_off = (ids // U) * np.uint64(packed_total) + uoff[ids % U]
        in_len = clen[ids % U].astype(np.uint64)
        out_off = ids * np.uint64(S)
        out_cap = np.full(NCH, S, np.uint64)
        b.out = torch.empty(NCH * S, dtype=torch.uint8, device=dev)
        b.bytes_in, b.bytes_out = int(in_len.sum()), NCH * S
    else:
        b.cin = raw.repeat(reps)
        in_off = ids * np.uint64(S)
        in_len = np.full(NCH, S, np.uint64)
        out_off = ids * np.uint64(stride_c)
        out_cap = np.full(NCH, stride_c, np.uint64)
        b.out = torch.empty(NCH * stride_c, dtype=torch.uint8, device=dev)
        b.bytes_in, b.bytes_out = NCH * S, None
    b.meta = torch.from_numpy(np.concatenate([in_off, in_len, out_off, out_cap, np.zeros(NCH, np.uint64)]).view(np.int64)).to(dev)
    return b


def batch_call(N, b, flags, op=None):
    mp = b.meta.data_ptr()
    op = (N.OP_DECOMPRESS if b.dec else N.OP_COMPRESS) if op is None else op
    return (b.codec, op, flags, b.NCH, b.cin.data_ptr(), mp, mp + 8 * b.NCH, b.out.data_ptr(), mp + 16 * b.NCH, mp + 24 * b.NCH, mp + 32 * b.NCH)


def verify_decoded(N, L, b, torch, dev):
    res = b.meta[4 * b.NCH:].cpu().numpy()
    assert (res == b.S).all(), "decode status/length mismatch: %s" % res[res != b.S][:8]
    mism = torch.zeros(1, dtype=torch.int64, device=dev)
    N.check(L.cj_bench_compare(b.out.data_ptr(), b.meta.data_ptr() + 16 * b.NCH, b.raw.data_ptr(), b.S, b.U, b.S, b.NCH, mism.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(mism.item()) == 0, "%d chunks decoded wrong" % int(mism.item())


def verify_compressed(N, L, b, torch, dev, np):
    """every compressed chunk of a compress batch: decoded with the GPU decoder (parity-tested against the CPU oracle in tests/)
    and compared with its input on the device, in slices of 3 x U chunks"""
    NCH, S, U = b.NCH, b.S, b.U
    clen = b.meta[4 * NCH:].cpu().numpy().view(np.int64)
    assert (clen > 0).all(), "compress status: %s" % clen[clen <= 0][:8]
    step = max(U, (24576 // U) * U)
    scratch = torch.empty(min(step, NCH) * S, dtype=torch.uint8, device=dev)
    for s0 in range(0, NCH, step):
        n = min(step, NCH - s0)
        ids = np.arange(s0, s0 + n, dtype=np.uint64)
        meta = torch.from_numpy(np.concatenate([ids * np.uint64(b.stride_c), clen[s0:s0 + n].astype(np.uint64), (ids - np.uint64(s0)) * np.uint64(S),
                                                np.full(n, S, np.uint64), np.zeros(n, np.uint64)]).view(np.int64)).to(dev)
        mp = meta.data_ptr()
        torch.cuda.synchronize()
        b.eng.batch_device(b.codec, N.OP_DECOMPRESS, 0, n, b.out.data_ptr(), mp, mp + 8 * n, scratch.data_ptr(), mp + 16 * n, mp + 24 * n, mp + 32 * n)
        b.eng.sync()
        res = meta[4 * n:].cpu().numpy()
        assert (res == S).all(), "round trip: decode status/length mismatch: %s" % res[res != S][:8]
        mism = torch.zeros(1, dtype=torch.int64, device=dev)
        N.check(L.cj_bench_compare(scratch.data_ptr(), mp + 16 * n, b.raw.data_ptr(), S, U, S, n, mism.data_ptr(), None))
        torch.cuda.synchronize()
        assert int(mism.item()) == 0, "%d compressed chunks do not decode to their input" % int(mism.item())
    return clen


def main():
    args = parse()
    in_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_torchrun and args.gpus > 1:
        relaunch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d — launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d), "
                         "or run `python bench.py --gpus %d` plainly and let it launch the ranks" % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist
    from cramjam_amd import _native as N

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("CJ_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()
        C.CDLL(None).fflush(None)          # RCCL's version banner sits in C stdio's buffer: out now, so that the JSON line is the last line
    L = N.lib()
    eng = N.Engine(local)

    mixed = args.workload == "mixed256k"
    S = args.chunk_bytes or (262144 if mixed else 65536)
    NCH = args.chunks or (16384 if mixed else (100_000 if world == 1 else 125_000))
    U = min(args.unique, NCH)
    if mixed:
        U = min(U, 2048)
    corpus_files = None
    if args.data == "corpus64k":
        if mixed or 65536 % S != 0:            # (experiments may cut the same bytes into 32 KiB chunks; the named workload is 64 KiB)
            raise SystemExit("bench.py: --data corpus64k is the 64 KiB single-codec workload")
        cc, corpus_files = corpus_chunks(S)
        U = min(U, len(cc))
    codec = N.CODEC_LZ4_BLOCK if args.codec == "lz4" else N.CODEC_SNAPPY_RAW
    dec = args.op in ("decompress", "roundtrip")
    mode_flag = {"auto": 0, "wave": N.FLAG_FORCE_WAVE_PER_CHUNK, "lane": N.FLAG_FORCE_LANE_PER_CHUNK,
                 "lds": N.FLAG_FORCE_LDS_PER_CHUNK}[args.lz4_mode] | (0x1000 if args.phase_profile else 0) | {"auto": 0, "fused": N.FLAG_FORCE_FUSED_PARSE, "kernel": N.FLAG_FORCE_PARSE_KERNEL}[args.parse] | (N.FLAG_BIG_CHUNKS if S > 65536 and args.lz4_mode == "auto" else 0) | (N.FLAG_CHUNKS_LE_16K if S <= 16384 else N.FLAG_CHUNKS_LE_32K if S <= 32768 else 0)      # (what the caller of a device batch knows about its chunks)

    # ---- workload: distinct synth-v1 chunk indices per rank (rank r generates indices r*U .. r*U+U-1) ----
    first_index = rank * U
    batches = []
    if mixed:
        # chunk i of this GPU: even -> LZ4 block, odd -> Snappy raw; one engine (stream) per codec
        eng2 = N.Engine(local)
        batches.append(build_batch(N, L, eng, dev, N.CODEC_LZ4_BLOCK, True, S, U, NCH - NCH // 2, first_index, args.compressor, torch, np))
        batches.append(build_batch(N, L, eng2, dev, N.CODEC_SNAPPY_RAW, True, S, U, NCH // 2, first_index + (1 << 32), args.compressor, torch, np))
    else:
        batches.append(build_batch(N, L, eng, dev, codec, dec, S, U, NCH, first_index, args.compressor, torch, np, args.data))
    b0 = batches[0]
    rt = None
    if args.op == "roundtrip":
        # configs[2]: compress the decoded batch again in the same step (the raw chunks are the compress input)
        rt = build_batch(N, L, eng, dev, codec, False, S, U, NCH, first_index, args.compressor, torch, np, args.data)
    torch.cuda.synchronize()

    calls = [(b.eng, batch_call(N, b, mode_flag)) for b in batches]
    # more than one batch in flight (single-codec workloads): extra engines with their own output and result buffers, same inputs
    lanes = [calls]
    extra = []
    for _ in range(1, max(1, args.in_flight)):
        if mixed:
            break
        e2 = N.Engine(local)
        out2 = torch.empty_like(b0.out)
        meta2 = b0.meta.clone()
        mp2 = meta2.data_ptr()
        a0 = calls[0][1]
        lanes.append([(e2, a0[:5] + (mp2, mp2 + 8 * NCH, out2.data_ptr(), mp2 + 16 * NCH, mp2 + 24 * NCH, mp2 + 32 * NCH))])
        extra.append((out2, meta2))
    torch.cuda.synchronize()

    single = len(lanes) == 1 and len(calls) == 1 and rt is None

    def run_steps(k):
        if single:
            return eng.batch_device_timed(*calls[0][1], k)       # K submissions on the engine stream, HIP events around them
        t = time.perf_counter()
        for i in range(k):
            for e_, a_ in lanes[i % len(lanes)]:
                e_.batch_device(*a_)                               # asynchronous: the codecs' streams run concurrently
            if rt is not None:
                eng.batch_device(*batch_call(N, rt, 0))
        for ln in lanes:
            for e_, _ in ln:
                e_.sync()
        return (time.perf_counter() - t) * 1e3 / k

    # ---- warmup, then EXACTLY K timed steps between barrier+synchronize on both sides ----
    if args.warmup > 0:
        run_steps(args.warmup)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = run_steps(args.steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    if args.phase_profile and rank == 0:
        ph = (C.c_ulonglong * 16)()
        L.cj_debug_lds_phase_cycles(ph, 1)
        nb = max(int(ph[5]), 1)
        if int(ph[5]):
            print("LDS decoder cycles/chunk: S0 %d  D1 %d  D2 %d  D3 %d  D4 %d  (blocks %d)  sub-marks 6.. %s" % (
                ph[0] // nb, ph[1] // nb, ph[2] // nb, ph[3] // nb, ph[4] // nb, nb, [round(ph[i] / nb, 2) for i in range(6, 16)]), file=sys.stderr)

    # ---- verify at full size: every chunk's result and every output byte ----
    bytes_in = sum(b.bytes_in for b in batches)
    if dec:
        for b in batches:
            if not args.experiment_no_verify:
                verify_decoded(N, L, b, torch, dev)
        for out2, meta2 in extra[:max(0, min(len(extra), args.steps + args.warmup - 1))]:
            res_k = meta2[4 * NCH:].cpu().numpy()
            assert (res_k == S).all()
            mism = torch.zeros(1, dtype=torch.int64, device=dev)
            N.check(L.cj_bench_compare(out2.data_ptr(), meta2.data_ptr() + 16 * NCH, b0.raw.data_ptr(), S, U, S, NCH, mism.data_ptr(), None))
            torch.cuda.synchronize()
            assert int(mism.item()) == 0
        bytes_out = sum(b.bytes_out for b in batches)
        ratio = bytes_out / bytes_in
        algo = bytes_in + bytes_out
    else:
        res = (b0.meta[4 * b0.NCH:].cpu().numpy().view(np.int64) if args.experiment_no_verify else verify_compressed(N, L, b0, torch, dev, np))
        bytes_out = int(res.sum())
        ratio = bytes_in / bytes_out
        algo = bytes_in + bytes_out
    if rt is not None:
        res = verify_compressed(N, L, rt, torch, dev, np)       # the compress half of the round trip
        algo += rt.bytes_in + int(res.sum())
    unc_bytes = sum(b.NCH for b in batches) * S

    from cramjam_amd.shard import aggregate
    wall_max, total_unc = aggregate(dist if use_dist else None, dev, wall, unc_bytes)

    cpu = None
    default_case = (not mixed and dec and rt is None and args.codec == "lz4" and S == 65536 and args.lz4_mode == "auto" and args.in_flight == 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, batches, args.op, rt)
    traffic = None
    kernels_ms = None
    if rank == 0 and world == 1 and (args.traffic == "on" or (args.traffic == "auto" and default_case)) and not os.environ.get("CJ_BENCH_CHILD"):
        traffic = measure_traffic()
        kernels_ms = measure_kernels()

    if rank == 0:
        achieved = algo / (kernel_ms * 1e-3)
        if mixed:
            metric = "uncompressed GB/s (mixed LZ4-block + Snappy-raw decomp, %d KiB chunks)" % (S // 1024)
            workload = "mixed: even chunks lz4-block / odd snappy-raw decompress, %d x %d B synth-v1 chunks per GPU, one stream per codec, device-resident" % (NCH, S)
            kernel = "lz4_parse_kernel+lz4_decode_lds2_kernel | snappy_parse_kernel+lz4_decode_lds2_kernel<snappy> (concurrent streams)"
        else:
            metric = ("uncompressed GB/s (LZ4-block decomp, 64 KiB chunks)" if (dec and rt is None and args.codec == "lz4" and S == 65536)
                      else "uncompressed GB/s (%s %s, %d B chunks)" % (args.codec, args.op, S))
            workload = "%s-block %s, %d x %d B %s chunks per GPU, device-resident" % (
                args.codec, args.op, NCH, S, "synth-v1" if corpus_files is None else "benchmark-corpus (%d unique full chunks of %d files — of the corpus's 1 413: the eight large files travel as 12-chunk samples —, tiled)" % (U, len(corpus_files)))
            if dec and args.codec == "lz4":
                kernel = {"auto": "lz4_parse_kernel+lz4_decode_lds2_kernel", "lds": "lz4_parse_kernel+lz4_decode_lds2_kernel",
                          "wave": "lz4_decode_kernel", "lane": "lz4_decode_lanes_kernel"}[args.lz4_mode]
            elif dec:
                kernel = "snappy_parse_kernel+lz4_decode_lds2_kernel<snappy>"
            else:
                kernel = encode_kernel_names(N, args.codec, NCH)
            if rt is not None:
                kernel += "+" + encode_kernel_names(N, args.codec, NCH)
        line = {
            "metric": metric,
            "value": total_unc / (wall_max / args.steps) / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if corpus_files is None else "reference benchmark corpus (tests/golden/corpus: %s), tiled" % ", ".join(corpus_files),
            "config": {"workload": workload, "chunks_per_gpu": NCH, "chunk_bytes": S, "unique_chunks": U, "ratio": round(ratio, 4),
                       "compressed_by": " | ".join(sorted({b.comp_name for b in batches if b.comp_name})) or None,
                       "batches_in_flight": len(lanes),
                       # (S > 64 KiB: what the engines hold for the big-chunk path — list, record areas of 1 MiB per planned chunk, summaries, slab tables — next to the bytes a step decodes)
                       **({"big_chunk_scratch_bytes": int(sum(L.cj_debug_big_scratch_bytes(b.eng.h) for b in batches)), "output_bytes_per_step": int(total_unc // world)} if S > 65536 and dec else {}),
                       "sharding": "chunk i -> gpu (i mod N), no collective; " + ("rank r generates synth-v1 indices r*%d .. r*%d+%d" % (U, U, U - 1) if corpus_files is None else "every rank tiles the same corpus chunks"),
                       "verified": "NOT VERIFIED (--experiment-no-verify): this line is INVALID as a result" if args.experiment_no_verify else
                                   ("all results + all output bytes compared on device" if dec and rt is None else
                                    "every compressed chunk decoded again by the GPU decoder and compared with its input on device"
                                    + ("; decode half: all results + all output bytes compared on device" if rt is not None else ""))},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic["total"] if traffic else None,
                         "traffic_detail": traffic, "kernel": kernel, "kernel_ms": kernel_ms, "kernels_ms": kernels_ms, "algorithmic_bytes_per_launch": algo},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def usable_cores():
    """threads worth starting on this host: logical CPUs, limited by the affinity mask and by a cgroup CPU quota (a
    container with 256 visible CPUs and a quota of 8 runs 256 threads at 8 cores' worth after the first burst)"""
    n = os.cpu_count() or 1
    note = "%d logical CPUs" % n
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, note = aff, note + ", affinity %d" % aff
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = (None if txt[0] == "max" else float(txt[0])), float(txt[1])
            else:
                quota, period = float(txt[0]), float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if quota <= 0 else quota
            if quota is not None:
                q = max(1, int(quota / period + 0.999))
                note += ", cgroup quota %.1f CPUs" % (quota / period)
                n = min(n, q)
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, note


def _cpu_legs(seconds, plan, n_of, run, check):
    """time every (kind, op, threads) leg of `plan` for its share of `seconds`: a calibration pass, then `reps` passes with one pool"""
    scale = seconds / sum(p[3] for p in plan)
    legs = []
    for kind, op, threads, share in plan:
        n1 = n_of(threads)
        t0 = time.perf_counter()
        run(op, threads, 1, n1)
        t1 = time.perf_counter() - t0
        reps = max(1, min(100000, int(share * scale / max(t1, 1e-6))))
        t0 = time.perf_counter()
        rc = run(op, threads, reps, n1)
        el = time.perf_counter() - t0
        assert rc == 0, "cpu leg %s failed" % kind
        check(kind, op, n1)
        legs.append((kind, threads, reps, n1, el))
    return legs


def cpu_baseline(args, batches, op, rt_batch=None):
    """The CPU side of the same work, on the U unique chunks of this run, ONE persistent thread pool per leg (oracle/synth_batch_oracle.c):
    the oracle's C restatement (kind "port") and the host's liblz4 / libsnappy when present — the C code the reference executes through
    lz4-sys, and the library its `snap` crate ports — each with all cores and with one.  decompress: the decoders over the same compressed
    chunks the GPU reads; compress: LZ4_compress_default / snappy_compress (and the port) over the same raw chunks; roundtrip: both halves
    (value = 1 / (1 / compress + 1 / decompress)); the mixed workload: each codec's half with its own library (value = bytes / summed time).
    Top level: the host library on all cores when present (else the port); `legs` lists everything."""
    import numpy as np
    import oracle
    OL = oracle.lib()
    cores, cores_note = usable_cores()
    seconds = args.cpu_seconds / max(1, (len(batches) if op != "roundtrip" else 2))
    all_legs, tops = [], []

    def one(b, enc):
        S, U = b.S, b.U
        lz4 = b.codec == 0
        raw_h = b.raw_h if b.raw_h is not None else b.raw.cpu().numpy()
        if enc:
            bound = (S + S // 255 + 16) if lz4 else (32 + S + S // 6)
            stride = (bound + 15) & ~15
            out = np.empty(U * stride, dtype=np.uint8)
            off = np.arange(U, dtype=np.uint64) * np.uint64(S)
            ln = np.full(U, S, np.uint64)
            src = raw_h
            plan = [("port", 1 if lz4 else 3, cores, 0.35), ("port", 1 if lz4 else 3, 1, 0.15)]
            if lz4 and OL.cjo_have_liblz4():
                plan += [("liblz4", 6, cores, 0.35), ("liblz4", 6, 1, 0.15)]
            if not lz4 and OL.cjo_have_libsnappy():
                plan += [("libsnappy", 7, cores, 0.35), ("libsnappy", 7, 1, 0.15)]
        else:
            stride = S
            out = np.empty(U * S, dtype=np.uint8)
            off = np.ascontiguousarray(b.uoff, dtype=np.uint64)
            ln = np.ascontiguousarray(b.clen, dtype=np.uint64)
            src = b.packed_h
            plan = [("port", 0 if lz4 else 2, cores, 0.35), ("port", 0 if lz4 else 2, 1, 0.15)]
            if lz4 and OL.cjo_have_liblz4():
                plan += [("liblz4", 4, cores, 0.35), ("liblz4", 4, 1, 0.15)]
            if not lz4 and OL.cjo_have_libsnappy():
                plan += [("libsnappy", 5, cores, 0.35), ("libsnappy", 5, 1, 0.15)]
        res = np.zeros(U, dtype=np.int64)

        def run(opc, threads, reps, n1):
            res[:] = 0
            return OL.cjo_batch_run_reps(opc, threads, reps, n1, src.ctypes.data, off.ctypes.data, ln.ctypes.data, out.ctypes.data, stride, res.ctypes.data)

        def check(kind, opc, n1):
            if enc:
                assert (res[:n1] > 0).all(), "cpu encoder (%s) failed" % kind
                for i in range(0, n1, max(1, n1 // 16)):          # a sample of its streams through the oracle's decoder
                    blk = out[i * stride:i * stride + int(res[i])].tobytes()
                    r, d = (oracle.lz4_decompress_raw(blk, S) if lz4 else oracle.snappy_decompress(blk))
                    assert r == S and d == raw_h[i * S:(i + 1) * S].tobytes(), "cpu encoder (%s) stream does not decode" % kind
            else:
                assert (res[:n1] == S).all() and (out[:n1 * S] == raw_h[:n1 * S]).all(), "cpu decoder (%s) disagrees with the generator" % kind
        legs = _cpu_legs(seconds, plan, lambda threads: U if threads > 1 else min(U, 256), run, check)
        what = ("%s %s" % ("lz4" if lz4 else "snappy", "compress" if enc else "decompress"))
        outl = [{"kind": kind, "what": what, "cores": threads, "value": reps * n1 * S / el / 1e9, "unit": "GB/s",
                 "sample": "%d passes x %d unique %d B chunks (same inputs as the GPU run), one thread pool, %.1f s" % (reps, n1, S, el)}
                for kind, threads, reps, n1, el in legs]
        best = next((g for g in outl if g["kind"] in ("liblz4", "libsnappy") and g["cores"] > 1), outl[0])
        return outl, best

    halves = []
    if op == "compress":
        halves = [(b, True) for b in batches]
    elif op == "decompress":
        halves = [(b, False) for b in batches]
    else:                                                     # roundtrip: the decode batch holds the compressed chunks, rt_batch the raw ones
        halves = [(rt_batch, True), (batches[0], False)]
    weights = []
    for b, enc in halves:
        legs, best = one(b, enc)
        all_legs += legs
        tops.append(best)
        weights.append(b.NCH * b.S)
    if op == "roundtrip":
        value = 1.0 / sum(1.0 / t["value"] for t in tops)
    else:
        value = sum(weights) / sum(w / t["value"] for w, t in zip(weights, tops))
    top = {"value": value, "unit": "GB/s", "cores": tops[0]["cores"], "kind": "+".join(sorted({t["kind"] for t in tops})),
           "sample": "; ".join("%s: %s" % (t["what"], t["sample"]) for t in tops), "host": cores_note, "legs": all_legs}
    if top["kind"] in ("liblz4", "libsnappy", "liblz4+libsnappy"):
        top["kind_note"] = "the host's C libraries of the codecs' own lineage (what the reference executes through lz4-sys / what its snap crate ports); SURVEY 8d kind: reference-lineage library, not the reference's Rust build"
    return top


def encode_kernel_names(N, codec, nch):
    """the kernel a compress batch runs as (lz4_encode.hip / snappy_encode.hip: one workgroup of two wavefronts per chunk) — the name rocprofv3 prints"""
    return "%s_encode_kernel<false, 2>" % codec


def measure_kernels():
    """average duration of every library kernel of a step, measured now by a child run of THIS command under
    rocprofv3 --kernel-trace --stats (3 timed steps + 1 warmup).  {kernel: ms per launch}; None when unavailable."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    base = [a for a in sys.argv[1:]]
    for flag in ("--steps", "--warmup", "--cpu-seconds", "--traffic"):
        while flag in base:
            i = base.index(flag)
            del base[i:i + 2]
    child = [sys.executable, os.path.abspath(__file__)] + base + ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--traffic", "off"]
    tmp = tempfile.mkdtemp(prefix="cj_kt_", dir="/tmp")
    try:
        env = dict(os.environ, CJ_BENCH_CHILD="1", TMPDIR="/tmp")
        r = subprocess.run([prof, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "--"] + child, cwd="/tmp", env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        out = {}
        for row in csv.DictReader(open(files[0])):
            k = row["Name"]
            if "cj::" not in k or "(anonymous" in k:
                continue
            out[k.split("(")[0].replace("void ", "")] = {"ms": float(row["AverageNs"]) / 1e6, "calls": int(row["Calls"])}
        return out
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic():
    """HBM bytes per step of THIS command, measured now: two child runs under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE —
    separate passes, the TCC block cannot hold both), 2 timed steps each, summed over every kernel of a step.
    Corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: KiB -> B; FETCH_SIZE x2 on gfx950 (wide coalesced
    reads are tallied at half their bytes); WRITE_SIZE as is (calibrated in round 1 on the decoder's exact output size).
    Returns None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    steps, warm = 2, 1
    base = [a for a in sys.argv[1:]]
    encodes = any(a in ("compress", "roundtrip") for a in base)
    for flag in ("--steps", "--warmup", "--cpu-seconds", "--traffic"):
        while flag in base:
            i = base.index(flag)
            del base[i:i + 2]
    child = [sys.executable, os.path.abspath(__file__)] + base + ["--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--traffic", "off"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="cj_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            env = dict(os.environ, CJ_BENCH_CHILD="1", TMPDIR="/tmp")
            r = subprocess.run([prof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    k = row["Kernel_Name"].split("(")[0]
                    if "cj::" not in k:                      # the library's kernels only (the bench utilities live in an anonymous namespace)
                        continue
                    if "encode" in k and not encodes:        # input preparation with the GPU encoder is not part of a step
                        continue
                    per.setdefault(k, []).append(float(row["Counter_Value"]))
            out[ctr] = per
    except (OSError, subprocess.SubprocessError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    launches = steps + warm
    detail, total = {}, 0.0
    for k in sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"])):
        f, w = out["FETCH_SIZE"].get(k, []), out["WRITE_SIZE"].get(k, [])
        # dispatches per step of this kernel = samples / launches (e.g. the routed wave kernel: one per step)
        rd = sum(f) / launches * 1024.0 * 2.0
        wr = sum(w) / launches * 1024.0
        detail[k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr}
        total += rd + wr
    return {"total": total, "unit": "bytes per step", "kernels": detail,
            "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child runs of this command (%d steps + %d warmup each); KiB->B, FETCH_SIZE x2 (gfx950)" % (steps, warm)}


if __name__ == "__main__":
    main()
