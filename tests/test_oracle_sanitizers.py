"""CPU-only: the oracle (the checker of the HIP engine) runs clean under AddressSanitizer and
UndefinedBehaviorSanitizer on every board size (oracle/sanitize_check.c)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "sanitize"], capture_output=True,
                         timeout=600)
    text = out.stdout.decode() + out.stderr.decode()
    assert out.returncode == 0, text[-3000:]
    assert "sanitize_check ok" in text and "runtime error" not in text and "AddressSanitizer" not in text
