"""Error-return paths of the C ABI that need no device, called through ctypes (run under the host-ASan build by
tools/asan_host.sh): every entry point with null / out-of-range arguments, reo_threshold over a range of n, the
create functions without a GPU, destroy(NULL)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import __graft_entry__ as ge
pkg = ge.load_pkg()
L = pkg._ffi.lib()
vp = ctypes.c_void_p
n_calls = 0
def call(name, *args):
    global n_calls
    n_calls += 1
    rc = getattr(L, name)(*args)
    L.reo_last_error()
    return rc
for n in range(2, 1501):
    call("reo_threshold", n, 0.01); call("reo_threshold", n, 0.2)
call("reo_threshold", 0, 0.01); call("reo_threshold", -5, 0.5); call("reo_threshold", 2, 2.0)
h = vp()
rc = call("reo_create", ctypes.byref(h), 0, 1)
has_gpu = rc == 0
if has_gpu:
    call("reo_destroy", h)
rc2 = call("reo_create", None, 0, 1)
assert rc2 != 0
h2 = vp()
call("reo_create_multi", ctypes.byref(h2), 0, 1)
if h2:
    call("reo_destroy", h2)
call("reo_create_multi", None, 0, 1)
call("reo_destroy", None)
null = vp()
z32 = np.zeros(64, dtype=np.int32); z64 = np.zeros(64, dtype=np.int64); zf = np.zeros(64)
p32 = z32.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)); pf = zf.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
# every entry point that takes a context, on a null context (argument checks come before any device work)
for name, (res, argt) in pkg._ffi.SIGNATURES.items() if hasattr(pkg._ffi, "SIGNATURES") else []:
    if not argt or argt[0] is not vp or name in ("reo_destroy",):
        continue
    args = [None] + [0 if a in (ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64) else (0.0 if a is ctypes.c_double else None) for a in argt[1:]]
    try:
        rc = call(name, *args)
        assert rc != 0 or name in ("reo_version",), name
    except ctypes.ArgumentError:
        pass
print("asan_host_calls: %d calls, device %s" % (n_calls, "present" if has_gpu else "absent"))
