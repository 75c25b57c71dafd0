"""ctypes binding of libbialign_hip.so (C ABI: include/bialign.h).

The library is built in-tree by ``bialign_amd.build`` (hipcc, gfx950).  There is
no fallback: if the shared object is missing or does not load, importing this
module raises -- the product never computes on the CPU.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# override: A/B builds of the engine (tools/exp_build.sh); a timing-experiment build is refused unless asked for by number
LIB_PATH = os.environ.get("BIALIGN_LIB_OVERRIDE") or os.path.join(HERE, "libbialign_hip.so")

ABI_VERSION = 9
RUN_FILL_ONLY = 1
RUN_ASYNC = 2
REC_AUTO, REC_AFFINE, REC_LINEAR = 0, 1, 2
MAX_SHIFT = 1024       # BIALIGN_MAX_SHIFT
MAX_SHIFT_TILED = 5    # BIALIGN_MAX_SHIFT_TILED: wider bands take the anti-diagonal path

E_INVALID, E_UNSUPPORTED, E_DEVICE, E_NOMEM, E_RANGE = -1, -2, -3, -4, -5

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


class Params(ctypes.Structure):
    _fields_ = [("gap_opening_cost", ctypes.c_int32), ("gap_cost", ctypes.c_int32),
                ("shift_cost", ctypes.c_int32), ("max_shift", ctypes.c_int32),
                ("recurrence", ctypes.c_int32), ("flags", ctypes.c_uint32)]


BATCH_SCORE_ONLY = 1  # BIALIGN_BATCH_SCORE_ONLY
BATCH_LEAN_TRACE = 2  # BIALIGN_BATCH_LEAN_TRACE


class Scoring(ctypes.Structure):
    _fields_ = [("k1", ctypes.c_int32), ("s1", c_i32p), ("k2", ctypes.c_int32), ("s2", c_i32p)]


class Pairs(ctypes.Structure):
    _fields_ = [("npairs", ctypes.c_int32), ("len_a", c_i32p), ("len_b", c_i32p),
                ("off_a", c_i64p), ("off_b", c_i64p), ("seq_a", c_u8p), ("cls_a", c_u8p),
                ("seq_b", c_u8p), ("cls_b", c_u8p), ("mu2_dense", c_i32p), ("mu2_off", c_i64p)]


class BatchInfo(ctypes.Structure):
    _fields_ = [("npairs", ctypes.c_int32), ("nchunks", ctypes.c_int32),
                ("affine", ctypes.c_int32), ("max_shift", ctypes.c_int32),
                ("cells", ctypes.c_int64), ("layer_bytes", ctypes.c_int64),
                ("hbm_layer_bytes", ctypes.c_int64), ("trace_bytes", ctypes.c_int64),
                ("storage", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class Timing(ctypes.Structure):
    _fields_ = [("fill_ms", ctypes.c_double), ("traceback_ms", ctypes.c_double),
                ("fill_launches", ctypes.c_int32), ("traceback_launches", ctypes.c_int32),
                ("waves_per_pair", ctypes.c_int32), ("cross_cu", ctypes.c_int32),
                ("recovered_runs", ctypes.c_int32), ("packed_records", ctypes.c_int32)]


#: every symbol include/bialign.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bialign_abi_version", ctypes.c_int, []),
    ("bialign_build_experiment", ctypes.c_int, []),
    ("bialign_device_count", ctypes.c_int, []),
    ("bialign_last_error", ctypes.c_char_p, []),
    ("bialign_engine_create", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("bialign_engine_destroy", None, [ctypes.c_void_p]),
    ("bialign_engine_trim", ctypes.c_int, [ctypes.c_void_p]),
    ("bialign_engine_reserve", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_double)]),
    ("bialign_batch_create", ctypes.c_int,
     [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.POINTER(Scoring), ctypes.POINTER(Pairs),
      ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    ("bialign_batch_destroy", None, [ctypes.c_void_p]),
    ("bialign_batch_get_info", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(BatchInfo)]),
    ("bialign_batch_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32]),
    ("bialign_batch_wait", ctypes.c_int, [ctypes.c_void_p]),
    ("bialign_batch_get_timing", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Timing)]),
    ("bialign_batch_get_scores", ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    ("bialign_batch_get_traces", ctypes.c_int, [ctypes.c_void_p, c_u8p, c_i64p, c_i32p, c_i32p]),
    ("bialign_batch_dump_layers", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_i32p]),
]


class BialignError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libbialign_hip error {code}: {message}")
        self.code = code
        self.message = message


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python -m bialign_amd.build). "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, also fatal
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.bialign_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libbialign_hip.so ABI {got} != expected {ABI_VERSION}; rebuild")
    exp = lib.bialign_build_experiment()
    if exp and os.environ.get("BIALIGN_ALLOW_EXPERIMENT_BUILD") != str(exp):
        # BIALIGN_LIB_OVERRIDE may point at a timing build (tools/exp_build.sh) whose results are wrong by construction:
        # never silently.  The timing tools say which experiment they expect.
        raise ImportError(f"{LIB_PATH} is a kernel timing experiment (BIALIGN_EXP={exp}), not a product build; "
                          f"set BIALIGN_ALLOW_EXPERIMENT_BUILD={exp} to time it")
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise BialignError(rc, lib.bialign_last_error().decode("utf-8", "replace"))
