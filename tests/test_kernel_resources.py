"""CPU: properties of the compiled step kernels that the GPU tests cannot see until they bite.

The big boards' step kernels issue their prefetches as inline-asm loads the compiler does not track.  Two
things then have to hold in the generated code, and did not always ("What the compiler cannot
see"):
  * no scratch: a spill of a register that an untracked load still has in flight stores garbage (seen: a
    `scratch_store` right behind the asm load, an iteration before the data arrived);
  * every asm load with a scalar base address starts with the wait states of the VALU-writes-SGPR -> VMEM
    hazard: the compiler pads its own instructions but does not look into inline asm (seen: the tile pointer
    reloaded from a spill lane by v_readlane_b32 right before the load -> stale base -> memory fault).
One hipcc -S of the engine (about a minute)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_scratch_and_hazard_padding(tmp_path):
    from twixt_for_open_spiel_amd import build
    asm = str(tmp_path / "engine.s")
    cmd = [build.hipcc()] + [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + [
        "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", asm, build.SRC]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    # ---- scratch of the kernels that prefetch through inline asm ----
    scratch, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    watched = {k: v for k, v in scratch.items() if re.search(r"step_window1_kernel|step_window_full_kernel|step_pair_kernel", k)}
    assert len(watched) >= 18, sorted(scratch)[:5]           # six board sizes x three kernels
    assert not {k: v for k, v in watched.items() if v}, "a kernel with untracked loads spills to scratch"
    # ---- every inline-asm load with a scalar base carries its wait states ----
    text = open(asm).read()
    blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, flags=re.S)
    loads = [b for b in blocks if re.search(r"global_load_dword\S*\s+\S+, v\d+, s\[", b)]
    assert len(loads) > 500
    bad = [b for b in loads if not re.search(r"s_nop 4\s*\n\s*global_load_dword", b)]  # (directly before the load)
    assert not bad, "asm load with a scalar base without hazard padding: %r" % bad[0]
