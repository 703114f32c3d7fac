"""Builds libtwixt_hip.so (the HIP engine + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "twixt_engine.hip")
DEPS = [os.path.join(PKG, "csrc", f) for f in sorted(os.listdir(os.path.join(PKG, "csrc")))] + [
    os.path.join(ROOT, "include", "twixt_hip.h")]
LIB = os.path.join(PKG, "libtwixt_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


# keep the ticket draw of the dynamic batch order a plain asynchronous atomic: the atomic
# optimizer would turn it into a wave reduction that waits for the result at once
# -cuid: a fixed compile-unit id.  By default hipcc hashes the source file's absolute path into the names of the
# anonymous-namespace kernels, so the same sources built in another directory gave another binary -- and
# bench.py believes a committed PMC traffic measurement only for the binary it was taken on (lib_stamp()).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-cuid=twixt_hip"]


def source_stamp():
    """sha256 over everything the library is built from (sources in name order + flags).
    Measurements that are only valid for one binary (profiles/*traffic*.json) carry it."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def lib_stamp():
    """sha256 of the built libtwixt_hip.so (None if it is not built)."""
    import hashlib
    if not os.path.exists(LIB):
        return None
    with open(LIB, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile if the library is missing or older than its sources.  Returns the path."""
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
