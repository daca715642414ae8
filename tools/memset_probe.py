import ctypes, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
p = ctypes.c_void_p()
n = 8 << 30
assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0
hip.hipDeviceSynchronize()
for rep in range(3):
    t0 = time.perf_counter(); rc = hip.hipMemset(p, 0, ctypes.c_size_t(n)); t1 = time.perf_counter()
    hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    print("hipMemset 8 GiB: call %.1f us, then sync %.1f us (rc %d)" % ((t1-t0)*1e6, (t2-t1)*1e6, rc))
for rep in range(3):
    t0 = time.perf_counter(); rc = hip.hipMemset(p, 0, ctypes.c_size_t(256)); t1 = time.perf_counter()
    hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    print("hipMemset 256 B: call %.1f us, then sync %.1f us (rc %d)" % ((t1-t0)*1e6, (t2-t1)*1e6, rc))
