"""CPU: the engine's address arithmetic, compiled from the product header for the host
(tests/layout_check.hip): tiled state layout, [slot][board] journal rows, backward batch order."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_layout_and_batch_order(tmp_path):
    exe = str(tmp_path / "layout_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "layout_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout + out.stderr
