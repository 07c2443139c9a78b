"""ctypes binding of the kernel-level C ABI (include/dacapo_ckks.h).  Device memory is handled through the
ABI's own dc_malloc/dc_memcpy_*: no torch types cross the boundary."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import LIB_PATH, LIB_PATH_GW

_lib = None
_libs: dict = {}


def lib_gw():
    """the generic-width build of the same sources (libSEAL_HEVM_gw.so: primes of 45..60 bits, mixed chains; csrc/modarith.hpp).  Device
    memory is the process's HIP runtime's either way, so DeviceBuffer pointers go to both libraries."""
    return lib(LIB_PATH_GW)


def lib(path=None):
    global _lib
    if path is not None and str(path) != str(LIB_PATH):
        if str(path) not in _libs:
            _libs[str(path)] = _bind(path)
        return _libs[str(path)]
    if _lib is None:
        _lib = _bind(LIB_PATH)
    return _lib


def loaded_libs():
    """the bound builds of the library, for runner.set_option (each build has its own option table)"""
    return ([_lib] if _lib is not None else []) + list(_libs.values())


def _bind(path):
    if True:  # (one indentation level kept from the single-library version)
        if not path.exists():
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HEVM hot path)")
        L = C.CDLL(str(path))
        vp, u64p, i32, lng = C.c_void_p, C.c_void_p, C.c_int, C.c_long
        L.dc_context_create.restype = vp
        L.dc_context_create.argtypes = [i32, i32, i32, vp]
        L.dc_context_create_hybrid.restype = vp
        L.dc_context_create_hybrid.argtypes = [i32, i32, i32, i32]
        L.dc_context_create_hybrid_primes.restype = vp
        L.dc_context_create_hybrid_primes.argtypes = [i32, vp, i32, i32, i32]
        L.dc_context_key_digits.argtypes = [vp]
        L.dc_context_max_level.argtypes = [vp]
        L.dc_context_destroy.argtypes = [vp]
        L.dc_context_logn.argtypes = [vp]
        L.dc_context_num_primes.argtypes = [vp]
        L.dc_context_primes.argtypes = [vp, vp]
        L.dc_context_roots.argtypes = [vp, vp]
        L.dc_malloc.restype = vp
        L.dc_malloc.argtypes = [C.c_size_t]
        L.dc_free.argtypes = [vp]
        L.dc_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.dc_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.dc_memset.argtypes = [vp, i32, C.c_size_t]
        L.dc_memcpy_d2d.argtypes = [vp, vp, C.c_size_t, vp]
        L.dc_stream_sync.argtypes = [vp]
        L.dc_set_device.argtypes = [i32]
        L.dc_event_create.restype = vp
        L.dc_event_destroy.argtypes = [vp]
        L.dc_event_record.argtypes = [vp, vp]
        L.dc_event_elapsed_ms.restype = C.c_float
        L.dc_event_elapsed_ms.argtypes = [vp, vp]
        for f in (L.dc_ntt_forward, L.dc_ntt_inverse):
            f.argtypes = [vp, u64p, lng, i32, vp, i32, i32, vp]
        L.dc_ntt_variant.argtypes = [vp, i32, i32, u64p, lng, i32, vp, i32, i32, vp]
        L.dc_ct_negate.argtypes = [vp, u64p, lng, u64p, lng, i32, vp]
        L.dc_ct_add.argtypes = [vp, u64p, lng, u64p, lng, u64p, lng, i32, vp]
        L.dc_ct_add_plain.argtypes = [vp, u64p, lng, u64p, lng, u64p, i32, vp]
        L.dc_ct_mul_plain.argtypes = [vp, u64p, lng, u64p, lng, u64p, i32, vp]
        L.dc_ct_mul_relin.argtypes = [vp, u64p, lng, u64p, lng, u64p, lng, u64p, i32, vp]
        L.dc_ct_rotate_hop.argtypes = [vp, u64p, lng, u64p, lng, C.c_uint32, u64p, i32, vp]
        L.dc_ct_rescale.argtypes = [vp, u64p, lng, u64p, lng, i32, vp]
        L.dc_ct_modswitch.argtypes = [vp, u64p, lng, u64p, lng, i32, i32, vp]
        L.dc_keyswitch.argtypes = [vp, u64p, lng, u64p, u64p, u64p, u64p, i32, vp]
        L.dc_galois_ntt.argtypes = [vp, u64p, lng, u64p, lng, C.c_uint32, i32, i32, vp]
        L.dc_poly_mul.argtypes = [vp, u64p, u64p, u64p, i32, vp]
        L.dc_poly_add.argtypes = [vp, u64p, u64p, u64p, i32, vp]
        L.dc_galois_elt_from_step.restype = C.c_uint32
        L.dc_galois_elt_from_step.argtypes = [vp, i32]
        from .runner import bind_options

        bind_options(L)
    return L


class DeviceBuffer:
    """A uint64 array in HBM owned through dc_malloc/dc_free."""

    def __init__(self, shape, dtype=np.uint64):
        self.shape = tuple(int(x) for x in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = lib().dc_malloc(max(self.nbytes, 16))

    @classmethod
    def from_host(cls, a: np.ndarray):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        lib().dc_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes)
        return d

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        lib().dc_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes)
        return out

    def at(self, elem_offset: int) -> int:
        return self.ptr + int(elem_offset) * self.dtype.itemsize

    def __del__(self):
        try:
            if self.ptr:
                lib().dc_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def read_device(ptr: int, shape, dtype=np.uint64) -> np.ndarray:
    """copy a raw device pointer range to a new host array"""
    out = np.empty(shape, dtype=dtype)
    lib().dc_memcpy_d2h(out.ctypes.data, ptr, out.nbytes)
    return out


class Context:
    def __init__(self, logN=15, num_primes=14, bit_size=60, primes=None, special=1, alpha=None):
        """special / alpha: the grouped-digit key-switching extension (dc_context_create_hybrid); 1 / 1 is SEAL's scheme"""
        narrow = bit_size != 60 or (primes is not None and any(int(p).bit_length() != 60 for p in primes))
        L = lib_gw() if narrow else lib()   # other prime widths: the generic-width build
        self.L = L
        arr = None
        if primes is not None:
            arr = (C.c_uint64 * len(primes))(*[int(p) for p in primes])
            num_primes = len(primes)
        alpha = special if alpha is None else alpha
        if (special, alpha) != (1, 1) and primes is not None:
            self.h = L.dc_context_create_hybrid_primes(logN, arr, num_primes, special, alpha)
        elif (special, alpha) != (1, 1):
            assert bit_size == 60
            self.h = L.dc_context_create_hybrid(logN, num_primes, special, alpha)
        else:
            self.h = L.dc_context_create(logN, num_primes, bit_size, arr)
        self.key_digits, self.max_level = int(L.dc_context_key_digits(self.h)), int(L.dc_context_max_level(self.h))
        self.logN, self.N, self.K = logN, 1 << logN, num_primes
        out = np.zeros(num_primes, dtype=np.uint64)
        L.dc_context_primes(self.h, out.ctypes.data)
        self.primes = [int(x) for x in out]
        L.dc_context_roots(self.h, out.ctypes.data)
        self.roots = [int(x) for x in out]

    def __del__(self):
        try:
            self.L.dc_context_destroy(self.h)
        except Exception:
            pass

    def sync(self, stream=None):
        lib().dc_stream_sync(stream)

    def ntt(self, buf: DeviceBuffer, count, inverse=False, prime_idx: DeviceBuffer | None = None, prime_base=0,
            prime_period=0, limb_stride=None, offset=0, stream=None, variant=None):
        """variant None: the library chooses by batch size; 0 / 1: the two-launch tiles / the single-crossing kernel (dc_ntt_variant)"""
        args = (buf.at(offset), self.N if limb_stride is None else limb_stride, count,
                prime_idx.ptr if prime_idx is not None else None, prime_base, prime_period, stream)
        if variant is None:
            (self.L.dc_ntt_inverse if inverse else self.L.dc_ntt_forward)(self.h, *args)
        else:
            self.L.dc_ntt_variant(self.h, variant, int(inverse), *args)

    def elt_from_step(self, step: int) -> int:
        return int(self.L.dc_galois_elt_from_step(self.h, step))
