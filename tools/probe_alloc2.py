"""hipMalloc by size (where does the slow path start?) and a large buffer made of 2 GiB physical chunks through the
virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess)."""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]


class Loc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class AllocFlags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class Prop(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", Loc), ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", AllocFlags)]


class Access(ctypes.Structure):
    _fields_ = [("location", Loc), ("flags", ctypes.c_int)]


def ms(f):
    t0 = time.perf_counter()
    r = f()
    return (time.perf_counter() - t0) * 1e3, r


def main():
    assert hip.hipSetDevice(0) == 0
    p = ctypes.c_void_p()
    hip.hipMalloc(ctypes.byref(p), 1 << 20)
    hip.hipFree(p)
    for gib in (1, 2, 3, 4, 5, 6, 8, 12):
        row = []
        for rep in range(3):
            q = ctypes.c_void_p()
            a, rc = ms(lambda: hip.hipMalloc(ctypes.byref(q), gib << 30))
            assert rc == 0
            f, _ = ms(lambda: hip.hipFree(q))
            row.append(f"{a:8.2f}/{f:5.2f}")
        print(f"hipMalloc {gib:2d} GiB (malloc/free ms): " + "  ".join(row), flush=True)
    prop = Prop()
    prop.type = 1           # hipMemAllocationTypePinned
    prop.location.type = 1  # hipMemLocationTypeDevice
    prop.location.id = 0
    gran = ctypes.c_size_t()
    rc = hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 1)   # recommended
    print("granularity rc", rc, gran.value)
    for total_gib, chunk_gib in ((16, 2), (16, 1), (16, 2), (32, 2)):
        total, chunk = total_gib << 30, chunk_gib << 30
        va = ctypes.c_void_p()
        t_res, rc = ms(lambda: hip.hipMemAddressReserve(ctypes.byref(va), ctypes.c_size_t(total), ctypes.c_size_t(0), ctypes.c_void_p(0), ctypes.c_ulonglong(0)))
        assert rc == 0, rc
        handles = []
        t_create = t_map = 0.0
        for i in range(total // chunk):
            h = ctypes.c_void_p()
            a, rc = ms(lambda: hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(chunk), ctypes.byref(prop), ctypes.c_ulonglong(0)))
            assert rc == 0, rc
            t_create += a
            a, rc = ms(lambda: hip.hipMemMap(ctypes.c_void_p(va.value + i * chunk), ctypes.c_size_t(chunk), ctypes.c_size_t(0), h, ctypes.c_ulonglong(0)))
            assert rc == 0, rc
            t_map += a
            handles.append(h)
        acc = Access()
        acc.location.type = 1
        acc.location.id = 0
        acc.flags = 3        # read-write
        t_acc, rc = ms(lambda: hip.hipMemSetAccess(va, ctypes.c_size_t(total), ctypes.byref(acc), ctypes.c_size_t(1)))
        assert rc == 0, rc
        t_set, _ = ms(lambda: (hip.hipMemset(va, 1, total), hip.hipDeviceSynchronize()))
        t_un, _ = ms(lambda: (hip.hipMemUnmap(va, ctypes.c_size_t(total)), [hip.hipMemRelease(h) for h in handles], hip.hipMemAddressFree(va, ctypes.c_size_t(total))))
        print(f"VMM {total_gib} GiB in {chunk_gib} GiB chunks: reserve {t_res:.2f} create {t_create:.2f} map {t_map:.2f} access {t_acc:.2f} memset {t_set:.2f} release {t_un:.2f} ms", flush=True)
    q = ctypes.c_void_p()
    a, rc = ms(lambda: hip.hipMalloc(ctypes.byref(q), 16 << 30))
    print(f"hipMalloc 16 GiB afterwards: {a:.2f} ms")


if __name__ == "__main__":
    main()
