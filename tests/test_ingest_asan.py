"""The host ingestion code under AddressSanitizer + UBSan (the GPU box cannot run sanitizers; the host part can be built
for the CPU alone): fuzzed FASTQ files -- ordinary, multi-line, blank lines, '+' in sequences, truncated, no final newline
-- through the raw-text windows and the host record scan with buffers of exactly the promised sizes.  No device needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "screencounter_amd", "csrc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("asan") / "ingest_asan_driver")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "ingest_asan_driver.cpp"), os.path.join(CSRC, "scg_ingest.cpp"), os.path.join(CSRC, "scg_fastq.cpp"),
           "-lz", "-ldl", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("host-only build of the ingestion code not possible here: " + r.stderr[-300:])
    return exe


@pytest.mark.parametrize("seed", [1, 2])
def test_host_ingestion_under_sanitizers(driver, tmp_path, seed):
    r = subprocess.run([driver, str(seed), "100", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout.startswith("ok:")


@pytest.fixture(scope="module")
def library_driver(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("asan_lib") / "library_asan_driver")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "library_asan_driver.cpp"), os.path.join(CSRC, "scg_library.cpp"), "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("host-only build of the library code not possible here: " + r.stderr[-300:])
    return exe


def test_index_builder_under_sanitizers(library_driver):
    """Pools of random sizes, lengths and budgets (ambiguity codes included): every entry must be reachable through every
    table the way the device looks for it, and the builder's threads must not touch anything they should not."""
    r = subprocess.run([library_driver, "3", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout.startswith("ok:")
