#!/usr/bin/env python3
"""bench.py — LZ4-block decompress throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over the whole device-resident batch: N chunks of 64 KiB
(synth-v1) compressed with the reference's C code family (system liblz4
LZ4_compress_default; falls back to this engine's own GPU encoder if the library is absent), already in
HBM when the timed region starts, decoded by one launch of lz4_decode_kernel into distinct outputs.
Weak scaling: every GPU owns `--chunks` chunks (independent units, no data-path collective).

Prints ONE JSON line on rank 0 (see the driver contract in the task description):
  value     = uncompressed GB/s over all ranks (sum of bytes / max wall time over ranks)
  roofline  = algorithmic bytes (compressed read + uncompressed written) per launch / mean launch
              duration measured with HIP events on the engine's stream, vs the 8 TB/s HBM peak
  cpu_baseline = the CPU oracle (scalar C restatement of liblz4's decoder) on this host's cores over a
              bounded sample of the same chunks (rank 0, N=1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--in-flight", type=int, default=1, help="batches in flight: step k is submitted to engine k mod N (own stream, outputs, results)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=100_000, help="chunks per GPU (BASELINE configs[1]: 100k x 64 KiB)")
    ap.add_argument("--chunk-bytes", type=int, default=65536)
    ap.add_argument("--unique", type=int, default=8192, help="distinct chunks generated; replicated device-side")
    ap.add_argument("--codec", default="lz4", choices=["lz4", "snappy"])
    ap.add_argument("--op", default="decompress", choices=["decompress", "compress"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compressor", default="auto", choices=["auto", "liblz4", "gpu"])
    ap.add_argument("--phase-profile", action="store_true", help="debug: per-phase cycle counters of the LDS decoder")
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from cramjam_amd import _native as N

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("CJ_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
    L = N.lib()
    eng = N.Engine(local)

    S, U, NCH = args.chunk_bytes, min(args.unique, args.chunks), args.chunks
    codec = N.CODEC_LZ4_BLOCK if args.codec == "lz4" else N.CODEC_SNAPPY_RAW
    dec = args.op == "decompress"

    # ---- workload: U unique synth-v1 chunks generated on the device ----
    raw = torch.empty(U * S, dtype=torch.uint8, device=dev)
    N.check(L.cj_bench_synth_v1(raw.data_ptr(), S, S, 0, U, 0x5EED, None))
    torch.cuda.synchronize()

    bound = L.cj_lz4_block_compress_bound(S, 0) if codec == N.CODEC_LZ4_BLOCK else L.cj_snappy_raw_max_compress_len(S)
    stride_c = (bound + 15) & ~15
    comp_name = None
    raw_h = None
    if dec or not args.no_cpu_baseline:
        raw_h = raw.cpu().numpy()
    if dec:
        lz4lib, comp_name = (None, None)
        if codec == N.CODEC_LZ4_BLOCK and args.compressor in ("auto", "liblz4"):
            lz4lib, comp_name = load_liblz4()
        if lz4lib is not None:
            from concurrent.futures import ThreadPoolExecutor
            comp_h = np.zeros(U * stride_c, dtype=np.uint8)
            clen = np.zeros(U, dtype=np.uint64)

            def work(i):
                r = lz4lib.LZ4_compress_default(raw_h.ctypes.data + i * S, comp_h.ctypes.data + i * stride_c, S, stride_c)
                assert r > 0
                clen[i] = r
            with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
                list(ex.map(work, range(U)))
        else:
            comp_name = "cramjam_amd GPU encoder (%s)" % args.codec
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
        packed = torch.from_numpy(packed_h).to(dev)
        reps = (NCH + U - 1) // U
        cin = packed.repeat(reps)                      # reps distinct copies in HBM
        ids = np.arange(NCH, dtype=np.uint64)
        in_off = (ids // U) * np.uint64(packed_total) + uoff[ids % U]
        in_len = clen[ids % U].astype(np.uint64)
        out_off = ids * np.uint64(S)
        out_cap = np.full(NCH, S, np.uint64)
        out = torch.empty(NCH * S, dtype=torch.uint8, device=dev)
        in_ptr = cin.data_ptr()
        bytes_in = int(in_len.sum()); bytes_out = NCH * S
    else:
        reps = (NCH + U - 1) // U
        cin = raw.repeat(reps)
        ids = np.arange(NCH, dtype=np.uint64)
        in_off = ids * np.uint64(S)
        in_len = np.full(NCH, S, np.uint64)
        out_off = ids * np.uint64(stride_c)
        out_cap = np.full(NCH, stride_c, np.uint64)
        out = torch.empty(NCH * stride_c, dtype=torch.uint8, device=dev)
        in_ptr = cin.data_ptr()
        bytes_in = NCH * S; bytes_out = None
    meta = torch.from_numpy(np.concatenate([in_off, in_len, out_off, out_cap, np.zeros(NCH, np.uint64)]).view(np.int64)).to(dev)
    mp = meta.data_ptr()
    mode_flag = {"auto": 0, "wave": N.FLAG_FORCE_WAVE_PER_CHUNK, "lane": N.FLAG_FORCE_LANE_PER_CHUNK,
                 "lds": N.FLAG_FORCE_LDS_PER_CHUNK}[args.lz4_mode] | (0x1000 if args.phase_profile else 0)
    a = (codec, N.OP_DECOMPRESS if dec else N.OP_COMPRESS, mode_flag, NCH, in_ptr, mp, mp + 8 * NCH, out.data_ptr(), mp + 16 * NCH,
         mp + 24 * NCH, mp + 32 * NCH)
    torch.cuda.synchronize()

    # more than one batch in flight: extra engines (streams) with their own output and result buffers, same inputs
    lanes = [(eng, a, out, meta)]
    for _ in range(1, max(1, args.in_flight)):
        e2 = N.Engine(local)
        out2 = torch.empty_like(out)
        meta2 = meta.clone()
        mp2 = meta2.data_ptr()
        lanes.append((e2, a[:4] + (in_ptr, mp2, mp2 + 8 * NCH, out2.data_ptr(), mp2 + 16 * NCH, mp2 + 24 * NCH, mp2 + 32 * NCH), out2, meta2))
    torch.cuda.synchronize()

    def run_steps(k):
        if len(lanes) == 1:
            return eng.batch_device_timed(*a, k)       # K launches on the engine stream, HIP events around them
        t = time.perf_counter()
        for i in range(k):
            e_, a_, _, _ = lanes[i % len(lanes)]
            e_.batch_device(*a_)
        for e_, _, _, _ in lanes:
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
        ph = (C.c_ulonglong * 8)()
        L.cj_debug_lds_phase_cycles(ph, 1)
        nb = max(int(ph[5]), 1)
        print("LDS decoder cycles/chunk: S0 %d  D1 %d  D2 %d  D3 %d  D4 %d  (blocks %d)" % (ph[0] // nb, ph[1] // nb, ph[2] // nb, ph[3] // nb, ph[4] // nb, nb), file=sys.stderr)

    # ---- verify at full size: every chunk's result and every output byte ----
    res = meta[4 * NCH:].cpu().numpy()
    if dec:
        for _, _, out_k, meta_k in lanes[:min(len(lanes), args.steps + args.warmup)]:
            res_k = meta_k[4 * NCH:].cpu().numpy()
            assert (res_k == S).all(), "decode status/length mismatch: %s" % res_k[res_k != S][:8]
            mism = torch.zeros(1, dtype=torch.int64, device=dev)
            N.check(L.cj_bench_compare(out_k.data_ptr(), mp + 16 * NCH, raw.data_ptr(), S, U, S, NCH, mism.data_ptr(), None))
            torch.cuda.synchronize()
            assert int(mism.item()) == 0, "%d chunks decoded wrong" % int(mism.item())
        ratio = bytes_out / bytes_in
    else:
        assert (res > 0).all()
        bytes_out = int(res.sum())
        ratio = bytes_in / bytes_out
    unc_bytes = NCH * S

    from cramjam_amd.shard import aggregate
    wall_max, total_unc = aggregate(dist if use_dist else None, dev, wall, unc_bytes)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and dec:
        cpu = cpu_baseline(args, codec, raw_h, S, U, packed_h, uoff, clen)

    if rank == 0:
        # HBM traffic from PMC counters cannot be sampled inside this process; it is measured with separate
        # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command and summarised by
        # tools/pmc_summary.py.  Reported only when the committed summary matches this exact workload.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc", "hbm_traffic_lz4_decode_100k_x_64k.json")))
            if dec and args.codec == "lz4" and NCH == 100_000 and S == 65536 and args.lz4_mode == "auto":
                traffic = tj["total_hbm_bytes_per_step"]
        except (OSError, ValueError, KeyError):
            pass
        algo = bytes_in + (bytes_out if bytes_out is not None else 0)
        achieved = algo / (kernel_ms * 1e-3)
        line = {
            "metric": "uncompressed GB/s (LZ4-block decomp, 64 KiB chunks)" if (dec and args.codec == "lz4" and S == 65536)
                      else "uncompressed GB/s (%s %s, %d B chunks)" % (args.codec, args.op, S),
            "value": total_unc / (wall_max / args.steps) / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s-block %s, %d x %d B synth-v1 chunks per GPU, device-resident" % (args.codec, args.op, NCH, S),
                       "chunks_per_gpu": NCH, "chunk_bytes": S, "unique_chunks": U, "ratio": round(ratio, 4),
                       "compressed_by": comp_name, "batches_in_flight": len(lanes), "sharding": "chunk i -> gpu (i mod N), no collective",
                       "verified": "all results + all output bytes compared on device"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel": {"auto": "lz4_parse_kernel+lz4_decode_lds2_kernel", "lds": "lz4_parse_kernel+lz4_decode_lds2_kernel", "wave": "lz4_decode_kernel",
                                    "lane": "lz4_decode_lanes_kernel"}[args.lz4_mode] if (dec and args.codec == "lz4") else ("snappy_parse_kernel+lz4_decode_lds2_kernel<snappy>" if (dec and args.lz4_mode in ("auto", "lds")) else "%s_%s_kernel" % (args.codec, "decode" if dec else "encode")),
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(args, codec, raw_h, S, U, packed_h, uoff, clen):
    """CPU oracle (oracle/: scalar C restatement of LZ4_decompress_safe / snap's decoder), all host cores,
    bounded sample: the U unique chunks, repeated until ~cpu-seconds elapsed."""
    import numpy as np
    import oracle
    OL = oracle.lib()
    cores = os.cpu_count() or 1
    threads = min(cores, 256)
    out = np.empty(U * S, dtype=np.uint8)
    res = np.zeros(U, dtype=np.int64)
    off = np.ascontiguousarray(uoff, dtype=np.uint64)
    ln = np.ascontiguousarray(clen, dtype=np.uint64)
    op = 0 if codec == 0 else 2
    done = 0
    t0 = time.perf_counter()
    while True:
        OL.cjo_batch_run(op, threads, U, packed_h.ctypes.data, off.ctypes.data, ln.ctypes.data, out.ctypes.data, S, res.ctypes.data)
        done += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or done >= 2000:
            break
    assert (res == S).all() and (out == raw_h).all(), "cpu oracle disagrees with generator"
    return {"value": done * U * S / el / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": "%d x %d unique %d B chunks (same inputs as the GPU run), %.1f s" % (done, U, S, el)}


if __name__ == "__main__":
    main()
