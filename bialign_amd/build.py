"""Build libbialign_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The kernels are templates on max_shift; each (max_shift, kind) slice is its own translation unit
(csrc/bialign_inst.hip with -DBIALIGN_TU_S / -DBIALIGN_TU_KIND), compiled in parallel, then linked
with the C-ABI unit (csrc/bialign_capi.hip) into one shared library.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
DEPS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "bialign.h"))
OUT = os.path.join(HERE, "libbialign_hip.so")
HOST_SRC = os.path.join(CSRC, "bialign_host.c")
HOST_OUT = os.path.join(HERE, "libbialign_host.so")
MAX_SHIFT = 5


def build_host(force=False):
    """Plain-C host helpers of the presentation layer (MEA fold); gcc, no GPU involved."""
    if force or not os.path.exists(HOST_OUT) or os.path.getmtime(HOST_OUT) < os.path.getmtime(HOST_SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", HOST_OUT, HOST_SRC], check=True)
    return HOST_OUT


def units():
    """(object name, source, extra flags), the slow ones (affine fill, wide bands) first."""
    out = [(f"inst_s{s}_k0.o", "bialign_inst.hip", [f"-DBIALIGN_TU_S={s}", "-DBIALIGN_TU_KIND=0"])
           for s in range(MAX_SHIFT, -1, -1)]
    out += [(f"inst_s{s}_k1.o", "bialign_inst.hip", [f"-DBIALIGN_TU_S={s}", "-DBIALIGN_TU_KIND=1"])
            for s in range(MAX_SHIFT, -1, -1)]
    out.append(("wide.o", "bialign_wide.hip", []))
    out.append(("capi.o", "bialign_capi.hip", []))
    return out


def build(force=False, verbose=False, out=OUT, defines=(), jobs=None):
    build_host(force)
    if not force and not defines and os.path.exists(out) and \
            all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = OBJ if out == OUT else out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + [f"-D{d}" for d in defines]
    if verbose:
        base.append("-Rpass-analysis=kernel-resource-usage")

    def compile_one(unit):
        obj, src, flags = unit
        subprocess.run(base + flags + ["-c", os.path.join(CSRC, src), "-o", os.path.join(objdir, obj)],
                       check=True)
        return os.path.join(objdir, obj)

    jobs = jobs or min(8, os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_one, units()))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
