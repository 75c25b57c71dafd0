"""Build libbialign_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bialign_capi.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "bialign_kernels.hpp"),
        os.path.join(os.path.dirname(HERE), "include", "bialign.h")]
OUT = os.path.join(HERE, "libbialign_hip.so")
HOST_SRC = os.path.join(HERE, "csrc", "bialign_host.c")
HOST_OUT = os.path.join(HERE, "libbialign_host.so")


def build_host(force=False):
    """Plain-C host helpers of the presentation layer (MEA fold); gcc, no GPU involved."""
    if force or not os.path.exists(HOST_OUT) or os.path.getmtime(HOST_OUT) < os.path.getmtime(HOST_SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", HOST_OUT, HOST_SRC], check=True)
    return HOST_OUT


def build(force=False, verbose=False):
    build_host(force)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
