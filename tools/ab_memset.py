"""Is plain streaming-write bandwidth allocation dependent on this box?  hipMalloc 83 GiB, time
hipMemsetD32Async with HIP events, free, repeat (with small allocations in between to shuffle)."""
import ctypes, sys
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
def ck(rc):
    if rc: raise RuntimeError(f"hip error {rc}")
GiB = 1 << 30
size = int(float(sys.argv[1]) * GiB) if len(sys.argv) > 1 else 83 * GiB
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
ck(hip.hipEventCreate(ctypes.byref(e0))); ck(hip.hipEventCreate(ctypes.byref(e1)))
junk = []
for cycle in range(8):
    p = ctypes.c_void_p()
    ck(hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size)))
    ts = []
    for _ in range(4):
        ck(hip.hipEventRecord(e0, None))
        ck(hip.hipMemsetD32Async(p, 7, ctypes.c_size_t(size // 4), None))
        ck(hip.hipEventRecord(e1, None)); ck(hip.hipEventSynchronize(e1))
        ms = ctypes.c_float(); ck(hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)); ts.append(ms.value)
    print(f"cycle {cycle}: ptr {p.value:#x}  memset ms " + " ".join(f"{t:.2f}" for t in ts) +
          f"   -> {size / min(ts) / 1e6:.0f} GB/s", flush=True)
    ck(hip.hipFree(p))
    q = ctypes.c_void_p(); ck(hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t((cycle + 1) * 3 * GiB))); junk.append(q)
