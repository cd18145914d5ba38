"""The per-lane DEFLATE decoder of the device inflate kernel (csrc/scg_inflate.h) compiled for the host and run against
zlib: valid streams of every block type and code shape decode identically; corrupted streams are rejected whenever zlib
rejects them, and give zlib's output when they stay valid; AddressSanitizer watches every access (the same code runs one
member per GPU lane, where an out-of-bounds access is a node-level fault).  No device needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("inflate") / "inflate_harness")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           os.path.join(ROOT, "tests", "inflate_harness.cpp"), "-lz", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_decoder_matches_zlib(harness, seed):
    r = subprocess.run([harness, str(seed), "150"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("ok:")
