"""Test infrastructure: the few HIP runtime calls a torch-free test process needs (device buffers, streams), through ctypes.
Used by tests/rccl_stub/run_cases.py, whose process must not import torch (torch brings its own librccl into the process)."""
import ctypes

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        _hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
        _hip.hipGetDevice.argtypes = [ctypes.POINTER(ctypes.c_int)]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


class DevBuf:
    """a device allocation; .ptr is the raw address"""

    def __init__(self, nbytes, fill=None, data=None):
        self.nbytes = max(int(nbytes), 1)
        p = ctypes.c_void_p()
        _ok(hip().hipMalloc(ctypes.byref(p), self.nbytes), "hipMalloc")
        self.ptr = p.value
        if data is not None:
            self.upload(data)
        elif fill is not None:
            _ok(hip().hipMemset(self.ptr, fill, self.nbytes), "hipMemset")

    def upload(self, data):
        data = bytes(data)
        assert len(data) <= self.nbytes
        if data:
            _ok(hip().hipMemcpy(self.ptr, data, len(data), 1), "hipMemcpy H2D")

    def download(self, nbytes=None):
        n = self.nbytes if nbytes is None else nbytes
        out = ctypes.create_string_buffer(max(n, 1))
        if n:
            _ok(hip().hipMemcpy(out, self.ptr, n, 2), "hipMemcpy D2H")
        return out.raw[:n]

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None


def device_synchronize():
    _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")


def current_device():
    d = ctypes.c_int(-1)
    _ok(hip().hipGetDevice(ctypes.byref(d)), "hipGetDevice")
    return d.value


class Stream:
    def __init__(self):
        p = ctypes.c_void_p()
        _ok(hip().hipStreamCreateWithFlags(ctypes.byref(p), 1), "hipStreamCreateWithFlags")   # hipStreamNonBlocking
        self.handle = p.value

    def synchronize(self):
        _ok(hip().hipStreamSynchronize(self.handle), "hipStreamSynchronize")

    def destroy(self):
        if self.handle:
            hip().hipStreamDestroy(self.handle)
            self.handle = None
