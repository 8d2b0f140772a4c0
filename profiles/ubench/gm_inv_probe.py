"""acf_hip_selftest_gradmag per binade: where gm_inv_fast differs from the IEEE form (development aid)."""
import ctypes as C
import sys
sys.path.insert(0, ".")
from acf_amd import capi
lib = capi.load()
ctx = C.c_void_p()
assert lib.acf_hip_create(0, None, C.byref(ctx)) == 0
tot = 0
for e in range(0, 255):
    lo, hi = e << 23, ((e + 1) << 23) - 1
    bad, first = C.c_uint64(0), C.c_uint32(0)
    assert lib.acf_hip_selftest_gradmag(ctx, lo, hi, C.byref(bad), C.byref(first)) == 0
    tot += bad.value
    if bad.value:
        import struct
        print("exp %3d (2^%d): %8d bad, first 0x%08x = %g" % (e, e - 127, bad.value, first.value, struct.unpack("f", struct.pack("I", first.value))[0]))
print("total", tot)
