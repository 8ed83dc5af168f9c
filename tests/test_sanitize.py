"""Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only;
GPU sanitizers are not available on this pool).  pfem_host.cpp needs no HIP: it is compiled here
with g++ -fsanitize=address,undefined together with a native driver that walks every host entry
point, including ragged / empty / out-of-contract inputs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_entry_points_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
           "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"),
           os.path.join(ROOT, "pfemfort_amd", "csrc", "pfem_host.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "host_sanitize: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
