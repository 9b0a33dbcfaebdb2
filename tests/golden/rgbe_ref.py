"""ctypes binding of oracle/_ref/librgbe_ref.so: the reference's own Common/rgbe.c behind oracle/rgbe_ref_shim.c.
TEST INFRASTRUCTURE ONLY.  Built by `make -C oracle ref` when /root/reference is present; the built library travels to
the GPU box with the tree (the reference's sources never do)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "_ref", "librgbe_ref.so")


def available():
    if not os.path.exists(LIB) and os.path.exists("/root/reference/Common/rgbe.c"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    return os.path.exists(LIB)


class RgbeRef:
    CAP = 1 << 24          # floats the harness is willing to let the reference write

    def __init__(self):
        if not available():
            raise FileNotFoundError(LIB)
        self.lib = C.CDLL(LIB)
        self.lib.rgbe_ref_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.lib.rgbe_ref_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        self.lib.rgbe_ref_encode.restype = C.c_size_t
        # the reference reports errors on stderr (rgbe.c:47-65); keep the test log readable
        self._null = None

    def decode(self, data):
        """(status, float32 [H, W, 3] or None); status 0 accepted, 1 header refused, 2 pixel data refused, 3 size not usable"""
        buf = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
        w, h = C.c_int(0), C.c_int(0)
        st = self.lib.rgbe_ref_decode(buf.ctypes.data, len(data), None, 0, C.byref(w), C.byref(h))
        if st != 0:
            return st, None
        if w.value * h.value * 3 > self.CAP:
            return 3, None
        img = np.zeros((h.value, w.value, 3), np.float32)
        st = self.lib.rgbe_ref_decode(buf.ctypes.data, len(data), img.ctypes.data, img.size, C.byref(w), C.byref(h))
        return st, (img if st == 0 else None)

    def encode(self, img):
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape[:2]
        n = self.lib.rgbe_ref_encode(img.ctypes.data, w, h, None, 0)
        out = np.zeros(max(n, 1), np.uint8)
        self.lib.rgbe_ref_encode(img.ctypes.data, w, h, out.ctypes.data, n)
        return out[:n].tobytes()
