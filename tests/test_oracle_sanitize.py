"""The oracle's C code under AddressSanitizer and UndefinedBehaviorSanitizer (host only): every entry
point on random, diverged and non-finite data, courses of one way-point, target indices beyond the
course (oracle/sanitize_check.c)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_is_clean_under_the_host_sanitizers():
    if shutil.which("gcc") is None or shutil.which("make") is None:
        pytest.skip("needs gcc and make")
    res = subprocess.run(["make", "-C", HERE, "-s", "sanitize"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "sanitize_check: rc 0" in res.stdout and "runtime error" not in res.stdout
