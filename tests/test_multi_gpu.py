"""Multi-GPU path: chunk i -> GPU i mod G, one engine per GPU, no collective.
* CPU: bench.py's launcher refuses configurations it cannot honour instead of silently running on one GPU
  (`--gpus N` without N devices, `--gpus` disagreeing with WORLD_SIZE).
* `-m gpu`: a batch sharded over devices [0, 1] gives per-chunk results identical to the 1-GPU run and to the oracle
  (skipped on a 1-GPU box); bench.py --gpus 2 prints n_gpus 2 with twice the chunks."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_more_gpus_than_visible():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = max(2, have + 1)
    r = _run(["--gpus", str(want), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs %d visible GPUs, found %d" % (want, have) in (r.stderr + r.stdout)


def test_bench_refuses_world_size_mismatch():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 1 but WORLD_SIZE=2" in (r.stderr + r.stdout)


def _device_count():
    from cramjam_amd import _native as N
    return N.lib().cj_device_count()


@pytest.mark.gpu
def test_batch_sharded_over_two_devices_matches_one_device_and_oracle():
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    import oracle
    from cramjam_amd import batch
    raws = [oracle.synth_v1(4096 + 257 * i, i) for i in range(61)] + [b"", b"howdy neighbor"]
    for enc, dec_many, one_dev in (
            (lambda r: oracle.lz4_compress_raw(r)[1], lambda blocks, devs: batch.lz4_decompress_blocks(blocks, [len(r) for r in raws], devices=devs), None),
            (lambda r: oracle.snappy_compress(r)[1], lambda blocks, devs: batch.snappy_decompress_raw_many(blocks, devices=devs), None)):
        blocks = [enc(r) for r in raws]
        res1, out1 = dec_many(blocks, [0])
        res2, out2 = dec_many(blocks, [0, 1])
        assert res1 == res2 == [len(r) for r in raws]
        assert out1 == out2 == raws
    # the second device alone, through its own engine
    res, outs = batch.lz4_decompress_blocks([oracle.lz4_compress_raw(r)[1] for r in raws], [len(r) for r in raws], devices=[1])
    assert outs == raws


@pytest.mark.gpu
def test_bench_two_gpus_line():
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--chunks", "8192", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["chunks_per_gpu"] == 8192 and line["scaling"] == "weak"
