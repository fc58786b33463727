#!/usr/bin/env python
"""Host-side cost of the HIP calls the streaming entry point makes per batch (tools/exp: measurement only).
ctypes on libamdhip64: hipMemcpyAsync H2D from page-locked memory at the three input sizes of one 128x416 window, a D2H of 48 B,
hipEventRecord, hipStreamWaitEvent, hipEventQuery, hipEventSynchronize on a completed event; issue time per call (the stream is
drained every 64 calls) and the rate of a copy-only stream."""
import ctypes
import time

L = ctypes.CDLL("libamdhip64.so")
vp, sz = ctypes.c_void_p, ctypes.c_size_t
L.hipMalloc.argtypes = [ctypes.POINTER(vp), sz]
L.hipHostMalloc.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_uint]
L.hipMemcpyAsync.argtypes = [vp, vp, sz, ctypes.c_int, vp]
L.hipMemcpy2DAsync.argtypes = [vp, sz, vp, sz, sz, sz, ctypes.c_int, vp]
L.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
L.hipStreamSynchronize.argtypes = [vp]
L.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
L.hipEventRecord.argtypes = [vp, vp]
L.hipStreamWaitEvent.argtypes = [vp, vp, ctypes.c_uint]
L.hipEventQuery.argtypes = [vp]
L.hipEventSynchronize.argtypes = [vp]


def ck(rc):
    assert rc == 0, rc


def main():
    s1, s2 = vp(), vp()
    ck(L.hipStreamCreateWithFlags(ctypes.byref(s1), 1)); ck(L.hipStreamCreateWithFlags(ctypes.byref(s2), 1))
    sizes = {"img 479 KB": 479232, "flow(2 planes) 852 KB": 851968, "seg 639 KB": 638976, "all 1.97 MB": 1970176, "64 windows 126 MB": 64 * 1970176}
    d, h = vp(), vp()
    ck(L.hipMalloc(ctypes.byref(d), 128 << 20)); ck(L.hipHostMalloc(ctypes.byref(h), 128 << 20, 0))
    ev = [vp() for _ in range(4)]
    for e in ev:
        ck(L.hipEventCreateWithFlags(ctypes.byref(e), 2))

    def timeit(name, fn, n=512, drain=64, stream=s1):
        fn(); ck(L.hipStreamSynchronize(stream))
        t_issue = 0.0
        t0 = time.perf_counter()
        for i in range(n):
            a = time.perf_counter()
            fn()
            t_issue += time.perf_counter() - a
            if i % drain == drain - 1:
                ck(L.hipStreamSynchronize(stream))
        ck(L.hipStreamSynchronize(stream))
        wall = time.perf_counter() - t0
        print("%-44s issue %7.2f us/call   wall %8.2f us/call" % (name, 1e6 * t_issue / n, 1e6 * wall / n), flush=True)

    for name, nb in sizes.items():
        n = 512 if nb < (8 << 20) else 24
        timeit("hipMemcpyAsync H2D pinned %s" % name, lambda: ck(L.hipMemcpyAsync(d, h, nb, 1, s1)), n=n, drain=8 if n < 100 else 64)
    timeit("hipMemcpy2DAsync H2D 64 x 852 KB of 1.7 MB", lambda: ck(L.hipMemcpy2DAsync(d, 1703936, h, 1703936, 851968, 64, 1, s1)), n=24, drain=8)
    timeit("hipMemcpyAsync D2H pinned 48 B", lambda: ck(L.hipMemcpyAsync(h, d, 48, 2, s1)))
    timeit("hipMemcpyAsync D2H pinned 3 KB", lambda: ck(L.hipMemcpyAsync(h, d, 3072, 2, s1)))
    timeit("hipEventRecord", lambda: ck(L.hipEventRecord(ev[0], s1)))
    timeit("hipEventRecord + hipStreamWaitEvent(other)", lambda: (ck(L.hipEventRecord(ev[1], s1)), ck(L.hipStreamWaitEvent(s2, ev[1], 0))))
    ck(L.hipStreamSynchronize(s2))
    ck(L.hipEventRecord(ev[2], s1)); ck(L.hipStreamSynchronize(s1))
    timeit("hipEventQuery (done)", lambda: L.hipEventQuery(ev[2]))
    timeit("hipEventSynchronize (done)", lambda: ck(L.hipEventSynchronize(ev[2])))
    # copy + record + sync per call: the hold = 0 pattern
    timeit("H2D 1.97 MB + record + eventSynchronize", lambda: (ck(L.hipMemcpyAsync(d, h, 1970176, 1, s1)), ck(L.hipEventRecord(ev[3], s1)), ck(L.hipEventSynchronize(ev[3]))))
    timeit("3 x H2D (img, flow, seg) + record + eventSynchronize", lambda: (ck(L.hipMemcpyAsync(d, h, 479232, 1, s1)), ck(L.hipMemcpyAsync(d, h, 851968, 1, s1)),
                                                                            ck(L.hipMemcpyAsync(d, h, 638976, 1, s1)), ck(L.hipEventRecord(ev[3], s1)), ck(L.hipEventSynchronize(ev[3]))))


if __name__ == "__main__":
    main()
