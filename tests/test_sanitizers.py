"""The two plain-C units -- the CPU oracle (test infrastructure) and the product's host helpers
(bialign_amd/csrc/bialign_host.c: MEA fold) -- rebuilt with AddressSanitizer + UBSan and driven by
the golden / host-mirror tests in a child interpreter.  CPU only (GPU sanitizers are not available)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc has no libasan.so")
def test_c_units_clean_under_asan_ubsan(tmp_path):
    flags = ["-O1", "-g", "-fPIC", "-shared", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    oracle_so, host_so = str(tmp_path / "liboracle_asan.so"), str(tmp_path / "libhost_asan.so")
    subprocess.run(["gcc", *flags, "-o", oracle_so, os.path.join(REPO, "oracle", "bialign_oracle.c")], check=True)
    subprocess.run(["gcc", *flags, "-o", host_so, os.path.join(REPO, "bialign_amd", "csrc", "bialign_host.c")], check=True)
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               BIALIGN_ORACLE_LIB_OVERRIDE=oracle_so, BIALIGN_HOST_LIB_OVERRIDE=host_so)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(REPO, "tests", "test_oracle_golden.py"),
                        os.path.join(REPO, "tests", "test_host_mirror.py")],
                       env=env, capture_output=True, text=True, cwd=REPO, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
