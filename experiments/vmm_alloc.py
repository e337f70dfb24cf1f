"""EXPERIMENT (round 6, not part of the package): device memory outside torch's caching allocator, through the HIP virtual-memory API.

Why: which PHYSICAL pages back a matrix decides how fast a random gather over it runs (profiles/r05_placement_cause.txt: the products
forward's dominant launch takes 18.0 or 19.5 ms on the same virtual addresses); torch's caching allocator hands out whatever block
history left it.  `vmm_tensor` creates the physical memory with hipMemCreate, reserves an aligned virtual range, maps it and wraps it as a
torch tensor (zero copy, through __cuda_array_interface__); the tensor owns the mapping and releases it when it is collected.
PyTorch is plumbing here: the HIP runtime it bundles is the one these calls go to (one runtime per process, see _lib.py)."""
import ctypes
import weakref

import torch

_hip = None


def _rt():
    global _hip
    if _hip is None:
        import glob
        import os
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*")) + ["libamdhip64.so"]
        err = None
        for c in cands:
            try:
                _hip = ctypes.CDLL(c)
                break
            except OSError as e:
                err = e
        if _hip is None:
            raise RuntimeError(f"glnn_amd.vmm: no HIP runtime ({err})")
    return _hip


class _Location(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class _AllocFlags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class _AllocProp(ctypes.Structure):       # hipMemAllocationProp
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", _Location), ("win32HandleMetaData", ctypes.c_void_p),
                ("allocFlags", _AllocFlags)]


class _AccessDesc(ctypes.Structure):      # hipMemAccessDesc
    _fields_ = [("location", _Location), ("flags", ctypes.c_int)]


_PINNED, _LOC_DEVICE, _RW = 1, 1, 3      # hipMemAllocationTypePinned, hipMemLocationTypeDevice, hipMemAccessFlagsProtReadWrite


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"glnn_amd.vmm: {what} failed (hipError {rc})")


def _prop(device):
    p = _AllocProp()
    p.type, p.requestedHandleType = _PINNED, 0
    p.location.type, p.location.id = _LOC_DEVICE, device
    return p


def granularity(device=0, recommended=True):
    g = ctypes.c_size_t(0)
    p = _prop(device)
    _check(_rt().hipMemGetAllocationGranularity(ctypes.byref(g), ctypes.byref(p), 1 if recommended else 0), "hipMemGetAllocationGranularity")
    return g.value


class _Mapping:
    """A reserved virtual range with physical handles mapped into it; exposes __cuda_array_interface__."""

    def __init__(self, nbytes, device, chunk_bytes, va_align):
        hip = _rt()
        gran = granularity(device, True)
        if chunk_bytes < 0:
            chunk_bytes = gran
        unit = max(gran, granularity(device, False))
        size = -(-nbytes // unit) * unit
        if chunk_bytes:
            chunk_bytes = -(-chunk_bytes // unit) * unit
            size = -(-size // chunk_bytes) * chunk_bytes
        self.size, self.device, self.handles, self.ptr = size, device, [], ctypes.c_void_p(0)
        _check(hip.hipMemAddressReserve(ctypes.byref(self.ptr), ctypes.c_size_t(size), ctypes.c_size_t(va_align), ctypes.c_void_p(0), ctypes.c_ulonglong(0)),
               "hipMemAddressReserve")
        prop = _prop(device)
        step = chunk_bytes or size
        try:
            for off in range(0, size, step):
                h = ctypes.c_void_p(0)
                _check(hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(step), ctypes.byref(prop), ctypes.c_ulonglong(0)), "hipMemCreate")
                self.handles.append(h)
                _check(hip.hipMemMap(ctypes.c_void_p(self.ptr.value + off), ctypes.c_size_t(step), ctypes.c_size_t(0), h, ctypes.c_ulonglong(0)), "hipMemMap")
            acc = _AccessDesc()
            acc.location.type, acc.location.id, acc.flags = _LOC_DEVICE, device, _RW
            _check(hip.hipMemSetAccess(self.ptr, ctypes.c_size_t(size), ctypes.byref(acc), ctypes.c_size_t(1)), "hipMemSetAccess")
        except Exception:
            self.release()
            raise
        self.mapped = True

    def release(self):
        hip = _rt()
        if self.ptr.value:
            torch.cuda.synchronize(self.device)
            if getattr(self, "mapped", False) or self.handles:
                hip.hipMemUnmap(self.ptr, ctypes.c_size_t(self.size))
            for h in self.handles:
                hip.hipMemRelease(h)
            hip.hipMemAddressFree(self.ptr, ctypes.c_size_t(self.size))
            self.ptr, self.handles = ctypes.c_void_p(0), []


class _Iface:
    def __init__(self, ptr, shape, owner):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}
        self._owner = owner


def _wrap(ptr, rows, cols, device, owner, release):
    t = torch.as_tensor(_Iface(ptr, (rows, cols), owner), device=torch.device(device))
    t._glnn_owner = owner                               # the mapping lives as long as the tensor (views keep the base tensor alive)
    weakref.finalize(t, release)
    return t


def vmm_tensor(rows, cols, device="cuda:0", chunk_bytes=0, va_align=1 << 30):
    """[rows, cols] float32 on `device`, backed by hipMemCreate'd physical memory: chunk_bytes = 0 one handle for everything, < 0 handles
    of the recommended granularity, > 0 handles of that size; the virtual range is aligned to va_align bytes."""
    idx = torch.device(device).index or 0
    m = _Mapping(rows * cols * 4, idx, chunk_bytes, va_align)
    return _wrap(m.ptr.value, rows, cols, device, m, m.release)


def hipmalloc_tensor(rows, cols, device="cuda:0"):
    """The same through a plain hipMalloc (no caching allocator in between)."""
    hip = _rt()
    torch.cuda.set_device(device)
    p = ctypes.c_void_p(0)
    _check(hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(rows * cols * 4)), "hipMalloc")

    def release(v=p.value):
        torch.cuda.synchronize()
        hip.hipFree(ctypes.c_void_p(v))
    return _wrap(p.value, rows, cols, device, None, release)
