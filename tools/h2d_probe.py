"""Developer probe: host->device copy rates on the box (pageable, registered, pinned staging) and the
cost of hipHostRegister, to choose how nbls_set_trace moves a 55 MB trace."""
import ctypes as C, time, numpy as np
hip = C.CDLL('/opt/rocm/lib/libamdhip64.so')
def chk(rc):
    assert rc == 0, rc
n = 864000 * 8
a = np.random.default_rng(0).standard_normal(n)
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), C.c_size_t(n * 8)))
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
for rep in range(3):
    t = time.perf_counter(); chk(hip.hipMemcpy(d, a.ctypes.data, n * 8, 1)); dt = time.perf_counter() - t
    print('pageable H2D %.2f ms  %.1f GB/s' % (dt * 1e3, n * 8 / dt / 1e9))
for rep in range(3):
    t = time.perf_counter(); chk(hip.hipHostRegister(a.ctypes.data, n * 8, 0)); t1 = time.perf_counter()
    chk(hip.hipMemcpy(d, a.ctypes.data, n * 8, 1)); t2 = time.perf_counter()
    chk(hip.hipHostUnregister(a.ctypes.data)); t3 = time.perf_counter()
    print('register %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms' % ((t1 - t) * 1e3, (t2 - t1) * 1e3, n * 8 / (t2 - t1) / 1e9, (t3 - t2) * 1e3))
p = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(p), n * 8, 0))
for rep in range(3):
    t = time.perf_counter(); C.memmove(p, a.ctypes.data, n * 8); t1 = time.perf_counter()
    chk(hip.hipMemcpy(d, p, n * 8, 1)); t2 = time.perf_counter()
    print('memmove to pinned %.2f ms (%.1f GB/s), pinned H2D %.2f ms (%.1f GB/s)' % ((t1 - t) * 1e3, n * 8 / (t1 - t) / 1e9, (t2 - t1) * 1e3, n * 8 / (t2 - t1) / 1e9))
out = np.empty(300000)
for rep in range(3):
    t = time.perf_counter(); chk(hip.hipMemcpy(out.ctypes.data, d, out.nbytes, 2)); dt = time.perf_counter() - t
    print('pageable D2H 2.4 MB %.3f ms' % (dt * 1e3))
for rep in range(3):
    t = time.perf_counter(); chk(hip.hipMemcpy(p, d, out.nbytes, 2)); dt = time.perf_counter() - t
    print('pinned D2H 2.4 MB %.3f ms' % (dt * 1e3))
t = time.perf_counter(); b = np.empty((8, 864000)); 
for i in range(8): b[i] = a[i * 864000:(i + 1) * 864000]
print('np.empty + row copies %.2f ms' % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter()
for i in range(8): b[i] = a[i * 864000:(i + 1) * 864000]
print('row copies again %.2f ms' % ((time.perf_counter() - t) * 1e3))
