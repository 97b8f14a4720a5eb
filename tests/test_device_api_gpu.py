"""cramjam_amd.batch.*_device: the device-resident batch behind a Python call (the reference's API is one Python call per buffer,
/root/reference/src/lz4.rs:78-131, src/snappy.rs:52-78; this is the same call for a batch that already sits in HBM).  Buffers are
torch tensors (any object with __cuda_array_interface__ or __dlpack__ works; the package itself never imports torch); every chunk
is compared with the oracle: 24 576 chunks decoded from tensors, compress round trips, DLPack-only objects, refusals.  The checks
run in a child process (tests/device_api_child.py) that imports torch first."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_resident_batches_from_torch_tensors():
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, os.path.join(HERE, "device_api_child.py")], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "device api: ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
