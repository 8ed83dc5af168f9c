"""Cost of hipMalloc / hipFree / first touch for multi-GB blocks (why a cold symbolic phase is slow)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def t(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
p = C.c_void_p()
for gb in (1, 4, 8):
    n = gb << 30
    for rep in range(3):
        a = t(lambda: hip.hipMalloc(C.byref(p), C.c_size_t(n)))
        b = t(lambda: (hip.hipMemset(p, 0, C.c_size_t(n)), hip.hipDeviceSynchronize()))
        c = t(lambda: (hip.hipMemset(p, 0, C.c_size_t(n)), hip.hipDeviceSynchronize()))
        d = t(lambda: hip.hipFree(p))
        print(f"{gb} GB rep {rep}: malloc {a:.1f} ms, first memset {b:.1f} ms, second memset {c:.1f} ms, free {d:.1f} ms")
