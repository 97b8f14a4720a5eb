"""Throughput of the device-resident batch through the Python entry (cramjam_amd.batch.lz4_decompress_blocks_device) with torch
tensors: 100 000 x 64 KiB synth-v1 chunks, liblz4-style streams from the oracle's encoder, timed with torch events around the call.
The README's example; the bench line (bench.py) measures the same batch through ctypes pointers."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle
from cramjam_amd import batch

U, n, S = 2048, 100000, 65536
raws = [oracle.synth_v1(S, i) for i in range(U)]
blobs = [oracle.lz4_compress_raw(r)[1] for r in raws]
ln_u = np.array([len(b) for b in blobs], np.uint64)
off_u = np.concatenate([[0], np.cumsum((ln_u + 15) & ~np.uint64(15))[:-1]]).astype(np.uint64)
buf = np.zeros(int(off_u[-1] + ln_u[-1]) + 64, np.uint8)
for k, b in enumerate(blobs): buf[int(off_u[k]):int(off_u[k]) + len(b)] = np.frombuffer(b, np.uint8)
dev = torch.device("cuda:0")
u_in = torch.from_numpy(buf).to(dev)
# every chunk of the batch at its own address (replicated on the device)
reps = (n + U - 1) // U
t_in = u_in.repeat(reps)
idx = np.arange(n)
off = (off_u[idx % U] + (idx // U).astype(np.uint64) * np.uint64(buf.size)).astype(np.uint64)
ln = ln_u[idx % U]
cap = np.full(n, S, np.uint64); out_off = (np.arange(n, dtype=np.uint64) * np.uint64(S))
t_out = torch.empty(n * S, dtype=torch.uint8, device=dev)
mk = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)
t_off, t_len, t_ooff, t_cap, t_res = mk(off), mk(ln), mk(out_off), mk(cap), torch.empty(n, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
torch.cuda.synchronize()
st = side.cuda_stream
for _ in range(3): batch.lz4_decompress_blocks_device(t_in, t_off, t_len, t_out, t_ooff, t_cap, result=t_res, stream=st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
with torch.cuda.stream(side):
    e0.record()
    for _ in range(K): batch.lz4_decompress_blocks_device(t_in, t_off, t_len, t_out, t_ooff, t_cap, result=t_res, stream=st, sync=False)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
assert bool((t_res == S).all())
got = t_out.view(n, S)[:U].cpu().numpy()
assert all(got[i].tobytes() == raws[i] for i in range(0, U, 97))
print("python entry, torch tensors: %d x %d B chunks, %.3f ms per batch, %.1f GB/s uncompressed" % (n, S, ms, n * S / ms / 1e6))
