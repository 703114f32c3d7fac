"""CPU-only: the host text renderer of the product (csrc/twixt_text.h) under ASan + UBSan
(tests/text_sanitize.cc): all board sizes, arbitrary link bits, edges and corners."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_text_renderer_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "text_sanitize")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-o", exe, os.path.join(HERE, "text_sanitize.cc")])
    out = subprocess.run([exe], capture_output=True, timeout=300)
    text = out.stdout.decode() + out.stderr.decode()
    assert out.returncode == 0, text[-3000:]
    assert "text_sanitize ok" in text and "runtime error" not in text
