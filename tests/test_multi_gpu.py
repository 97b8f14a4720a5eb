"""Multi-GPU path: chunk i -> GPU i mod G, one engine per GPU, no collective.
* CPU: bench.py's launcher refuses configurations it cannot honour instead of silently running on one GPU
  (`--gpus N` without N devices, `--gpus` disagreeing with WORLD_SIZE).
* `-m gpu`: a batch sharded over devices [0, 1] gives per-chunk results identical to the 1-GPU run and to the oracle
  (skipped on a 1-GPU box); bench.py --gpus 2 prints n_gpus 2 with twice the chunks; on ANY box: the bench's RCCL control plane with a
  world of one rank, and a batch sharded over two engines that live on device 0."""
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


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_control_plane_on_rccl_with_one_rank():
    """the multi-GPU bench's control plane — init_process_group("nccl", device_id), barrier, the device all-reduces of
    cramjam_amd/shard.py:aggregate — runs on RCCL with a world of ONE rank, so it is exercised on a 1-GPU box too"""
    env = {"CJ_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = _run(["--gpus", "1", "--chunks", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--traffic", "off"], env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith('{"metric"'), r.stdout[-1500:]          # the JSON line is the LAST line (the driver reads it)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["chunks_per_gpu"] == 8192 and line["value"] > 0
    assert "compared on device" in line["config"]["verified"]


@pytest.mark.gpu
def test_batch_over_two_engines_on_one_device_matches_oracle():
    """cramjam_amd/batch.py shards a host batch over its engines from one host thread each; with devices [0, 0] two threads drive
    the same engine concurrently (its host staging is serialised by the engine's mutex) and two DISTINCT engines on device 0 are
    driven through batch._run's own code path — per-chunk results equal the oracle's, in the caller's order"""
    import oracle
    from cramjam_amd import batch, _native as N
    raws = [oracle.synth_v1(3000 + 331 * i, 100 + i) for i in range(83)] + [b"", b"howdy neighbor", bytes(70000)]
    blocks = [oracle.lz4_compress_raw(r)[1] for r in raws]
    res, outs = batch.lz4_decompress_blocks(blocks, [len(r) for r in raws], devices=[0, 0])
    assert res == [len(r) for r in raws] and outs == raws
    # two separate engines on the same GPU (what a second GPU's engine would be), chunk i -> engine i mod 2
    saved = dict(batch._engines)
    try:
        batch._engines.clear()
        batch._engines[0] = N.Engine(0)
        batch._engines["second"] = N.Engine(0)
        res2, outs2 = batch._run(N.CODEC_LZ4_BLOCK, N.OP_DECOMPRESS, 0, blocks, [len(r) for r in raws], [0, "second"])
        assert res2 == res and outs2 == raws
        sblocks = [oracle.snappy_compress(r)[1] for r in raws]
        res3, outs3 = batch._run(N.CODEC_SNAPPY_RAW, N.OP_DECOMPRESS, 0, sblocks, [len(r) for r in raws], [0, "second"])
        assert res3 == res and outs3 == raws
    finally:
        for k in ("second",):
            e = batch._engines.pop(k, None)
            if e is not None:
                e.close()
        batch._engines.clear()
        batch._engines.update(saved)
