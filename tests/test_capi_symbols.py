"""The C-ABI library builds, loads and exports every symbol include/bialign.h
declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    with open(os.path.join(REPO, "include", "bialign.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(bialign_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported():
    from bialign_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    names = declared_functions()
    assert len(names) >= 13
    for name in names:
        assert hasattr(lib, name), f"{name} declared in bialign.h but not exported"


def test_binding_table_covers_header():
    from bialign_amd import _lib
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == declared_functions()
    assert _lib.lib.bialign_abi_version() == _lib.ABI_VERSION
    with open(os.path.join(REPO, "include", "bialign.h")) as fh:
        assert f"#define BIALIGN_ABI_VERSION {_lib.ABI_VERSION}" in fh.read()


def test_no_device_is_a_loud_error():
    """Without a GPU every entry point fails with a message -- never a CPU fallback."""
    from bialign_amd import _lib
    if _lib.lib.bialign_device_count() > 0:
        return  # on the GPU box
    h = ctypes.c_void_p()
    rc = _lib.lib.bialign_engine_create(0, ctypes.byref(h))
    assert rc < 0 and _lib.lib.bialign_last_error()


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(REPO, "bialign_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                with open(os.path.join(root, f)) as fh:
                    src = fh.read()
                assert "oracle" not in src.lower() or f == "verify.py", os.path.join(root, f)


def test_timing_experiment_builds_are_refused(tmp_path):
    """BIALIGN_LIB_OVERRIDE may name another build of the engine; one compiled as a kernel timing experiment
    (-DBIALIGN_EXP=n: wrong results by construction) must not load silently.  A stub with the header's symbols
    stands in for such a build (a real one takes a minute of hipcc)."""
    import subprocess
    import sys
    from bialign_amd import _lib
    body = "".join(f"long {n}(void) {{ return {_lib.ABI_VERSION if n == 'bialign_abi_version' else (3 if n == 'bialign_build_experiment' else 0)}; }}\n"
                   for n, _, _ in _lib.SYMBOLS)
    src, so = tmp_path / "stub.c", tmp_path / "libstub.so"
    src.write_text(body)
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    env = dict(os.environ, BIALIGN_LIB_OVERRIDE=str(so), PYTHONPATH=REPO)
    env.pop("BIALIGN_ALLOW_EXPERIMENT_BUILD", None)
    r = subprocess.run([sys.executable, "-c", "import bialign_amd._lib"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "timing experiment (BIALIGN_EXP=3)" in r.stderr
    r = subprocess.run([sys.executable, "-c", "import bialign_amd._lib"], env=dict(env, BIALIGN_ALLOW_EXPERIMENT_BUILD="3"),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _lib.lib.bialign_build_experiment() == 0   # the shipped library is a product build
