"""Mirror of the reference's python/hecate/hecate/runner.py (class HEVM, runner.py:174-271) bound to this repo's
libSEAL_HEVM.so.  Same method names, argument meaning and call sequence; the differences are noted inline.

    hevm = HEVM(path)                    # runner.py:175-202  (create_context on first use + initFullVM)
    hevm.load(cst_path, hevm_path)       # runner.py:205-221  (load + preprocess)
    hevm.setInput(i, data)               # runner.py:227-231  (encrypt)
    hevm.run()                           # runner.py:223-225  (run + printMem, the timed region)
    res = hevm.getOutput()               # runner.py:239-254  (decrypt_result per result)
"""
from __future__ import annotations

import contextlib
import ctypes
import re
import weakref
from pathlib import Path

import numpy as np

from . import LIB_PATH

lw = None


def reinit_lw():  # runner.py:73-117
    global lw
    if lw is None:
        lw = bind_vm_lib(LIB_PATH)
    return lw


_vm_libs: dict = {}


def bind_vm_lib(path):
    """the 18 reference symbols + the hevm_* extensions of one build of the library (the default build, or libSEAL_HEVM_gw.so for VMs on
    chains with primes narrower than 60 bits)"""
    if str(path) in _vm_libs:
        return _vm_libs[str(path)]
    if not Path(path).exists():
        raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build() (no CPU fallback exists)")
    L = _vm_libs[str(path)] = ctypes.CDLL(str(path))
    L.initFullVM.argtypes = [ctypes.c_char_p, ctypes.c_bool]
    L.initFullVM.restype = ctypes.c_void_p
    L.initClientVM.argtypes = [ctypes.c_char_p]
    L.initClientVM.restype = ctypes.c_void_p
    L.initServerVM.argtypes = [ctypes.c_char_p]
    L.initServerVM.restype = ctypes.c_void_p
    L.create_context.argtypes = [ctypes.c_char_p]
    L.load.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    L.loadClient.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.getArgLen.argtypes = [ctypes.c_void_p]
    L.getArgLen.restype = ctypes.c_int64
    L.getResLen.argtypes = [ctypes.c_void_p]
    L.getResLen.restype = ctypes.c_int64
    L.encrypt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    L.decrypt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    L.decrypt_result.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    L.getResIdx.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.getResIdx.restype = ctypes.c_int64
    L.getCtxt.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.getCtxt.restype = ctypes.c_void_p
    L.preprocess.argtypes = [ctypes.c_void_p]
    L.run.argtypes = [ctypes.c_void_p]
    L.setDebug.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    L.setToGPU.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    L.printMem.argtypes = [ctypes.c_void_p]
    # extensions of include/hevm_abi.h
    L.hevm_init_fresh.argtypes = [ctypes.c_int, ctypes.c_int]
    L.hevm_init_fresh.restype = ctypes.c_void_p
    L.hevm_init_fresh_primes.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    L.hevm_init_fresh_primes.restype = ctypes.c_void_p
    L.has_test_hooks = hasattr(L, "hevm_init_seeded")  # the *_hooks.so builds only (csrc/test_hooks.hip; DACAPO_AMD_HOOKS=1)
    if L.has_test_hooks:
        L.hevm_init_seeded.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        L.hevm_init_seeded.restype = ctypes.c_void_p
        L.hevm_init_seeded_primes.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_uint64]
        L.hevm_init_seeded_primes.restype = ctypes.c_void_p
        L.hevm_secret_key.argtypes = [ctypes.c_void_p]
        L.hevm_secret_key.restype = ctypes.c_void_p
        L.hevm_test_zero_encryption.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    L.hevm_context.argtypes = [ctypes.c_void_p]
    L.hevm_context.restype = ctypes.c_void_p
    for f in (L.hevm_relin_key, L.hevm_public_key):
        f.argtypes = [ctypes.c_void_p]
        f.restype = ctypes.c_void_p
    L.hevm_key_buffers.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hevm_key_digest.argtypes = [ctypes.c_void_p]
    L.hevm_key_digest.restype = ctypes.c_uint64
    L.hevm_keys_replaced.argtypes = [ctypes.c_void_p]
    L.hevm_galois_key.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    L.hevm_galois_key.restype = ctypes.c_void_p
    L.hevm_plain.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]
    L.hevm_plain.restype = ctypes.c_void_p
    L.hevm_plain_special.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.hevm_plain_special.restype = ctypes.c_void_p
    L.hevm_load_mem.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
    L.hevm_last_run_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int64)]
    L.hevm_plan_lazy_groups.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64]
    L.hevm_plan_lazy_groups.restype = ctypes.c_int64
    L.hevm_set_streams.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.hevm_select_stream.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.hevm_last_run_bootstrap_seconds.argtypes = [ctypes.c_void_p]
    L.hevm_last_run_bootstrap_seconds.restype = ctypes.c_double
    L.hevm_plaintext_bytes.argtypes = [ctypes.c_void_p]
    L.hevm_plaintext_bytes.restype = ctypes.c_uint64
    L.hevm_destroy.argtypes = [ctypes.c_void_p]
    L.hevm_destroy.restype = None
    L.hevm_add_rotation_keys.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    L.hevm_save_ctxt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
    L.hevm_load_ctxt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
    bind_options(L)
    for name, value in _cli_options.items():  # `--opt` pairs given before this build was loaded (each build has its own option table)
        L.hevm_set_option(name.encode(), int(value))
    return L


_cli_options: dict = {}


def bind_options(L):
    L.hevm_set_option.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    L.hevm_get_option.argtypes = [ctypes.c_char_p]
    L.hevm_get_option.restype = ctypes.c_longlong
    L.hevm_reset_options.argtypes = []
    return L


def _option_libs():
    """every loaded build of the library (the default one and, when a narrow-prime context has been made, the generic-width one): each has
    its own option table"""
    from . import lowlevel

    reinit_lw()
    libs, seen = [], set()
    for L in list(_vm_libs.values()) + lowlevel.loaded_libs():  # (the VM binding and the kernel-level binding of one build: one table)
        if L._name not in seen:
            seen.add(L._name)
            libs.append(L)
    return libs


def set_option(name: str, value: int):
    """extension: hevm_set_option (include/hevm_abi.h; the table of names is dacapo_amd/csrc/options.hpp) on every loaded build"""
    for L in _option_libs():
        L.hevm_set_option(name.encode(), int(value))


def get_option(name: str) -> int:
    reinit_lw()
    return int(lw.hevm_get_option(name.encode()))


def apply_cli_options(argv):
    """tools: strip `--opt name=value` pairs from an argument list and set them (measurement sweeps without rebuilding or forking)"""
    out, i = [], 0
    while i < len(argv):
        if argv[i] == "--opt" and i + 1 < len(argv):
            name, _, value = argv[i + 1].partition("=")
            set_option(name, int(value, 0))
            _cli_options[name] = int(value, 0)  # ... and on builds loaded later (bind_vm_lib)
            i += 2
        else:
            out.append(argv[i])
            i += 1
    return out


@contextlib.contextmanager
def options(**kw):
    """set options for the duration of a with-block and put the previous values back: VM options are read when a VM is created, launch
    shapes at every launch -- `with options(plan=0): HEVM(...)`, `with options(sum_pair_min_wgs=0): hevm.run()`"""
    libs = _option_libs()  # each build has its own table: saved and restored per build
    for L in libs:
        bind_options(L)
    old = [{k: int(L.hevm_get_option(k.encode())) for k in kw} for L in libs]
    try:
        for k, v in kw.items():
            set_option(k, v)
        yield
    finally:
        for L, saved in zip(libs, old):
            for k, v in saved.items():
                L.hevm_set_option(k.encode(), v)


class hevm_ctxt(ctypes.Structure):  # include/hevm_abi.h
    _fields_ = [("data", ctypes.c_void_p), ("poly_stride", ctypes.c_int64), ("level", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("scale", ctypes.c_double)]


run_library = "SEAL"
run_hardware = "GPU"


def setLibnHW(argv=None):  # runner.py:123-171: only the SEAL-compatible ABI exists here, always on the MI355X
    return


_live_vms: dict = {}  # VM handle -> weak reference to its HEVM object, for every VM whose device state has not been released


def close_all():
    """extension: hevm_destroy every VM this process has created and not closed -- also those whose Python object is gone (the reference's
    runner never frees a VM).  Test suites call it between modules: one module's VMs must not crowd the next module's out of HBM.
    Surviving HEVM objects lose their handle (vm = None), so a later call on one fails in Python instead of touching freed memory."""
    for handle, (ref, lib_) in list(_live_vms.items()):
        lib_.hevm_destroy(handle)
        obj = ref()
        if obj is not None:
            obj.vm = None
        _live_vms.pop(handle, None)


class HEVM:
    def __init__(self, path=str((Path.home() / ".hevm" / "seal").absolute()), option="full", seed=None, logN=0, num_primes=0,
                 ks_special=1, ks_alpha=None, vm_options=None, primes=None, fresh=False):
        """fresh (extension): parameters and keys generated in HBM from the OS's randomness, nothing on disk (hevm_init_fresh); `primes`
        implies it.  seed (TEST HOOK, the *_hooks.so builds only): the same with every key expanded from the seed -- reproducible, not secret.
        ks_special / ks_alpha (extension): grouped-digit hybrid key switching -- the last ks_special primes are special, a digit is
        ks_alpha (default ks_special) data primes.  1 / 1 = the reference's SEAL scheme.  vm_options: further VM options of
        csrc/options.hpp (plan, plan_graph, secret_hw, logn, primes, ...), in force while this VM is created; the previous values are put
        back afterwards (a VM keeps what it was created with).  primes (extension, seeded VMs): an explicit chain, e.g. a HEaaN-style
        mixed one (60-bit base and special primes around 51-bit rescale primes); a chain with primes narrower than 60 bits -- given here or
        through vm_options["prime_bits"] -- runs on the generic-width build of the same sources (libSEAL_HEVM_gw.so)."""
        reinit_lw()
        from . import LIB_PATH_GW

        narrow = (vm_options or {}).get("prime_bits", 60) != 60 or (primes is not None and any(int(q).bit_length() != 60 for q in primes))
        self.lw = lw_ = bind_vm_lib(LIB_PATH_GW) if narrow else lw
        self.option = option
        self.slots = None
        opts = dict(vm_options or {})
        if ks_special != 1 or (ks_alpha or 1) != 1:
            opts.update(ks_special=ks_special, ks_alpha=ks_alpha or ks_special)
        with options(**opts):
            if seed is not None and not lw_.has_test_hooks:
                raise RuntimeError("HEVM(seed=...) needs the hooks build of the library (DACAPO_AMD_HOOKS=1 before importing dacapo_amd: "
                                   "tests/conftest.py); the release build generates keys from the OS's randomness only -- HEVM(fresh=True)")
            if seed is not None and primes is not None:
                arr = (ctypes.c_uint64 * len(primes))(*[int(q) for q in primes])
                self.vm = lw_.hevm_init_seeded_primes(logN, arr, len(primes), seed)
            elif seed is not None:  # test hook: keys generated in HBM from a seed, nothing on disk
                self.vm = lw_.hevm_init_seeded(logN, num_primes, seed)
            elif primes is not None:
                arr = (ctypes.c_uint64 * len(primes))(*[int(q) for q in primes])
                self.vm = lw_.hevm_init_fresh_primes(logN, arr, len(primes))
            elif fresh:  # extension: keys generated in HBM, nothing on disk
                self.vm = lw_.hevm_init_fresh(logN, num_primes)
            else:
                if not Path(path).is_dir():  # runner.py:185-192 (the reference also waits for a key press)
                    Path(path).mkdir(parents=True)
                    lw_.create_context(path.encode("utf-8"))
                if option == "full":
                    self.vm = lw_.initFullVM(path.encode("utf-8"), True)
                elif option == "client":
                    self.vm = lw_.initClientVM(path.encode("utf-8"))
                elif option == "server":
                    self.vm = lw_.initServerVM(path.encode("utf-8"))
                else:
                    raise ValueError(option)
        from . import lowlevel

        _live_vms[self.vm] = (weakref.ref(self), lw_)
        L = lowlevel.lib(LIB_PATH_GW) if narrow else lowlevel.lib()
        self.ctx_handle = lw_.hevm_context(self.vm)
        self.logN = L.dc_context_logn(self.ctx_handle)
        self.K = L.dc_context_num_primes(self.ctx_handle)
        self.key_digits, self.max_level = int(L.dc_context_key_digits(self.ctx_handle)), int(L.dc_context_max_level(self.ctx_handle))
        self.N = 1 << self.logN
        self.slots = self.N >> 1

    def load(self, const_path, hevm_path, preprocess=True):
        if not Path(const_path).is_file():
            raise Exception(f"No file exists in const_path {const_path}")
        if not Path(hevm_path).is_file():
            raise Exception(f"No file exists in hevm_path {hevm_path}")
        if self.option in ("full", "server"):
            self.lw.load(self.vm, str(const_path).encode("utf-8"), str(hevm_path).encode("utf-8"))
        elif self.option == "client":
            self.lw.loadClient(self.vm, str(hevm_path).encode("utf-8"))  # the reference passes const_path here (upstream bug)
        if preprocess:
            self.lw.preprocess(self.vm)
        else:
            raise Exception("Not implemented in SEAL_HEVM")
        self.arglen = self.lw.getArgLen(self.vm)
        self.reslen = self.lw.getResLen(self.vm)
        self.hevm_path = str(hevm_path)

    def load_mem(self, cst: bytes, hevm: bytes, preprocess=True):
        """extension: load from memory images (no temp files)"""
        self.lw.hevm_load_mem(self.vm, cst, len(cst), hevm, len(hevm))
        if preprocess:
            self.lw.preprocess(self.vm)
        self.arglen = self.lw.getArgLen(self.vm)
        self.reslen = self.lw.getResLen(self.vm)
        self.hevm_path = "<memory>"

    def set_streams(self, n):
        """extension: n independent ciphertext streams through the same program (call before load)"""
        self.lw.hevm_set_streams(self.vm, n)

    def select_stream(self, s):
        self.lw.hevm_select_stream(self.vm, s)

    def run(self):
        self.lw.run(self.vm)
        self.lw.printMem(self.vm)

    def setInput(self, i, data):
        if not isinstance(data, np.ndarray):
            data = np.array(data, dtype=np.float64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        carr = data.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self.lw.encrypt(self.vm, i, carr, len(data))

    def setDebug(self, enable):
        self.lw.setDebug(self.vm, enable)

    def setToGPU(self, ongpu):
        self.lw.setToGPU(self.vm, ongpu)

    def getOutput(self):
        result = np.zeros((self.reslen, self.slots), dtype=np.float64)  # reference: (reslen, 1 << 14)
        data = np.zeros(self.slots, dtype=np.float64)
        for i in range(self.reslen):
            carr = data.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            self.lw.decrypt_result(self.vm, i, carr)
            result[i] = data
        return result

    def keyBuffers(self):
        """[(device pointer, 64-bit words)] of the key material in canonical order (hevm_key_buffers)"""
        n = self.lw.hevm_key_buffers(self.vm, None, None, 0)
        ptrs, words = (ctypes.c_void_p * n)(), (ctypes.c_uint64 * n)()
        self.lw.hevm_key_buffers(self.vm, ptrs, words, n)
        return [(int(ptrs[i] or 0), int(words[i])) for i in range(n)]

    def keyDigest(self) -> int:
        return int(self.lw.hevm_key_digest(self.vm))

    def keysReplaced(self):
        self.lw.hevm_keys_replaced(self.vm)

    def close(self):
        """extension: return this VM's HBM (hevm_destroy).  The reference's runner never frees its VM; neither does this class unless asked."""
        if getattr(self, "vm", None) and self.vm in _live_vms:  # (not already released by close_all)
            self.lw.hevm_destroy(self.vm)
            _live_vms.pop(self.vm, None)
        self.vm = None

    def plaintextBytes(self) -> int:
        """extension: HBM held for the program's plaintexts (pre-encoded pool, or constants + window with option online_encode = 1)"""
        return int(self.lw.hevm_plaintext_bytes(self.vm))

    def addRotationKeys(self, offsets):
        """extension: direct Galois keys for these slot offsets (create_galois_keys(steps) in SEAL; HEAAN_HEVM.cpp:58-64's key list)"""
        arr = (ctypes.c_int64 * len(offsets))(*[int(o) for o in offsets])
        self.lw.hevm_add_rotation_keys(self.vm, arr, len(offsets))

    def saveCtxt(self, reg: int, path):
        """extension: seal::Ciphertext::save of a cipher register (SEAL 4.0 bytes) -- what a client / server pair exchanges"""
        self.lw.hevm_save_ctxt(self.vm, reg, str(path).encode("utf-8"))

    def loadCtxt(self, reg: int, path):
        self.lw.hevm_load_ctxt(self.vm, reg, str(path).encode("utf-8"))

    def getResIdx(self, i: int) -> int:
        return int(self.lw.getResIdx(self.vm, i))

    def getCtxt(self, reg: int) -> hevm_ctxt:
        return hevm_ctxt.from_address(self.lw.getCtxt(self.vm, reg))

    def lazy_groups(self):
        """option hyb_lazy_sum: the plan's lazy sums as lists of rotate-instruction indices (hevm_plan_lazy_groups); [] before the first run"""
        n = self.lw.hevm_plan_lazy_groups(self.vm, None, 0)
        if n <= 0:
            return []
        buf = (ctypes.c_int32 * n)()
        self.lw.hevm_plan_lazy_groups(self.vm, buf, n)
        out, i = [], 0
        while i < n:
            out.append(list(buf[i + 1 : i + 1 + buf[i]]))
            i += 1 + buf[i]
        return out

    def stats(self):
        counts = (ctypes.c_int64 * 11)()
        ks, ntt = ctypes.c_int64(), ctypes.c_int64()
        self.lw.hevm_last_run_stats(self.vm, counts, ctypes.byref(ks), ctypes.byref(ntt))
        return {"op_counts": list(counts), "keyswitches": ks.value, "ntts": ntt.value,
                "bootstrap_s": float(self.lw.hevm_last_run_bootstrap_seconds(self.vm))}

    def printer(self, latency, rms, mem_usage=0.0):  # runner.py:256-271
        bench = re.search(r"optimized/(.*)/(.*)\.(.*)\._", self.hevm_path)
        print("======================================")
        print("---------------Option-----------------")
        if bench:
            print("compiler:", bench.group(1))
            print("benchname:", bench.group(2))
            print("waterline:", bench.group(3))
        print("library:", run_library)
        print("device:", run_hardware)
        print("---------------Result-----------------")
        print("latency:", latency)
        print("rms:", rms)
        print("======================================")
        print()
