"""How long do hipMalloc / hipFree of large blocks take on this stack?  (ctypes on libamdhip64, no torch)
usage: probe_alloc.py            -> table of sizes x repetitions, fresh process"""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]


def t(f):
    t0 = time.perf_counter()
    r = f()
    return (time.perf_counter() - t0) * 1e3, r


def main():
    assert hip.hipSetDevice(0) == 0
    p = ctypes.c_void_p()
    ms, _ = t(lambda: hip.hipMalloc(ctypes.byref(p), 1 << 20))
    print(f"first hipMalloc (context): {ms:.1f} ms")
    hip.hipFree(p)
    for mb in (64, 512, 2048, 8192, 16384):
        row = []
        for rep in range(4):
            q = ctypes.c_void_p()
            a, rc = t(lambda: hip.hipMalloc(ctypes.byref(q), mb << 20))
            assert rc == 0, rc
            m, _ = t(lambda: (hip.hipMemset(q, 0, mb << 20), hip.hipDeviceSynchronize()))
            f, _ = t(lambda: hip.hipFree(q))
            row.append(f"malloc {a:8.2f} memset {m:7.2f} free {f:8.2f}")
        print(f"{mb:6d} MiB: " + " | ".join(row))
    # many live blocks, then a churn like a pattern build: allocate 8 x 2 GiB, free them in order, allocate again
    blocks = []
    a, _ = t(lambda: [blocks.append(ctypes.c_void_p()) or hip.hipMalloc(ctypes.byref(blocks[-1]), 2 << 30) for _ in range(8)])
    f, _ = t(lambda: [hip.hipFree(b) for b in blocks])
    print(f"8 x 2 GiB: malloc {a:.1f} ms, free {f:.1f} ms")
    blocks = []
    a, _ = t(lambda: [blocks.append(ctypes.c_void_p()) or hip.hipMalloc(ctypes.byref(blocks[-1]), 2 << 30) for _ in range(8)])
    f, _ = t(lambda: [hip.hipFree(b) for b in blocks])
    print(f"again:     malloc {a:.1f} ms, free {f:.1f} ms")


if __name__ == "__main__":
    main()
