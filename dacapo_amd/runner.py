"""Mirror of the reference's python/hecate/hecate/runner.py (class HEVM, runner.py:174-271) bound to this repo's
libSEAL_HEVM.so.  Same method names, argument meaning and call sequence; the differences are noted inline.

    hevm = HEVM(path)                    # runner.py:175-202  (create_context on first use + initFullVM)
    hevm.load(cst_path, hevm_path)       # runner.py:205-221  (load + preprocess)
    hevm.setInput(i, data)               # runner.py:227-231  (encrypt)
    hevm.run()                           # runner.py:223-225  (run + printMem, the timed region)
    res = hevm.getOutput()               # runner.py:239-254  (decrypt_result per result)
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

import numpy as np

from . import LIB_PATH

lw = None


def reinit_lw():  # runner.py:73-117
    global lw
    if lw is not None:
        return lw
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (no CPU fallback exists)")
    lw = ctypes.CDLL(str(LIB_PATH))
    lw.initFullVM.argtypes = [ctypes.c_char_p, ctypes.c_bool]
    lw.initFullVM.restype = ctypes.c_void_p
    lw.initClientVM.argtypes = [ctypes.c_char_p]
    lw.initClientVM.restype = ctypes.c_void_p
    lw.initServerVM.argtypes = [ctypes.c_char_p]
    lw.initServerVM.restype = ctypes.c_void_p
    lw.create_context.argtypes = [ctypes.c_char_p]
    lw.load.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    lw.loadClient.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lw.getArgLen.argtypes = [ctypes.c_void_p]
    lw.getArgLen.restype = ctypes.c_int64
    lw.getResLen.argtypes = [ctypes.c_void_p]
    lw.getResLen.restype = ctypes.c_int64
    lw.encrypt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lw.decrypt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    lw.decrypt_result.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    lw.getResIdx.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    lw.getResIdx.restype = ctypes.c_int64
    lw.getCtxt.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    lw.getCtxt.restype = ctypes.c_void_p
    lw.preprocess.argtypes = [ctypes.c_void_p]
    lw.run.argtypes = [ctypes.c_void_p]
    lw.setDebug.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    lw.setToGPU.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    lw.printMem.argtypes = [ctypes.c_void_p]
    # extensions of include/hevm_abi.h
    lw.hevm_init_seeded.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
    lw.hevm_init_seeded.restype = ctypes.c_void_p
    lw.hevm_context.argtypes = [ctypes.c_void_p]
    lw.hevm_context.restype = ctypes.c_void_p
    for f in (lw.hevm_relin_key, lw.hevm_secret_key, lw.hevm_public_key):
        f.argtypes = [ctypes.c_void_p]
        f.restype = ctypes.c_void_p
    lw.hevm_key_buffers.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lw.hevm_key_digest.argtypes = [ctypes.c_void_p]
    lw.hevm_key_digest.restype = ctypes.c_uint64
    lw.hevm_keys_replaced.argtypes = [ctypes.c_void_p]
    lw.hevm_galois_key.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lw.hevm_galois_key.restype = ctypes.c_void_p
    lw.hevm_plain.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]
    lw.hevm_plain.restype = ctypes.c_void_p
    lw.hevm_load_mem.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
    lw.hevm_last_run_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int64)]
    lw.hevm_set_streams.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lw.hevm_select_stream.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lw.hevm_last_run_bootstrap_seconds.argtypes = [ctypes.c_void_p]
    lw.hevm_last_run_bootstrap_seconds.restype = ctypes.c_double
    lw.hevm_plaintext_bytes.argtypes = [ctypes.c_void_p]
    lw.hevm_plaintext_bytes.restype = ctypes.c_uint64
    lw.hevm_destroy.argtypes = [ctypes.c_void_p]
    lw.hevm_destroy.restype = None
    lw.hevm_add_rotation_keys.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    lw.hevm_test_zero_encryption.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    lw.hevm_save_ctxt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
    lw.hevm_load_ctxt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
    return lw


class hevm_ctxt(ctypes.Structure):  # include/hevm_abi.h
    _fields_ = [("data", ctypes.c_void_p), ("poly_stride", ctypes.c_int64), ("level", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("scale", ctypes.c_double)]


run_library = "SEAL"
run_hardware = "GPU"


def setLibnHW(argv=None):  # runner.py:123-171: only the SEAL-compatible ABI exists here, always on the MI355X
    return


_live_vms: dict = {}  # id(HEVM object) -> VM handle, for every VM whose device state has not been released (close / close_all)


def close_all():
    """extension: hevm_destroy every VM this process has created and not closed -- also those whose Python object is gone (the reference's
    runner never frees a VM).  Test suites call it between modules: one module's VMs must not crowd the next module's out of HBM."""
    for key, handle in list(_live_vms.items()):
        lw.hevm_destroy(handle)
        _live_vms.pop(key, None)


class HEVM:
    def __init__(self, path=str((Path.home() / ".hevm" / "seal").absolute()), option="full", seed=None, logN=0, num_primes=0,
                 ks_special=1, ks_alpha=None):
        """ks_special / ks_alpha (extension, seeded VMs): grouped-digit hybrid key switching -- the last ks_special primes are special, a
        digit is ks_alpha (default ks_special) data primes.  1 / 1 = the reference's SEAL scheme."""
        reinit_lw()
        self.option = option
        self.slots = None
        if seed is not None:  # extension: keys generated in HBM, nothing on disk
            env = {}
            if ks_special != 1 or (ks_alpha or 1) != 1:
                env = {"DACAPO_HEVM_KS_SPECIAL": str(ks_special), "DACAPO_HEVM_KS_ALPHA": str(ks_alpha or ks_special)}
            os.environ.update(env)
            try:
                self.vm = lw.hevm_init_seeded(logN, num_primes, seed)
            finally:
                for k in env:
                    os.environ.pop(k)
        else:
            if not Path(path).is_dir():  # runner.py:185-192 (the reference also waits for a key press)
                Path(path).mkdir(parents=True)
                lw.create_context(path.encode("utf-8"))
            if option == "full":
                self.vm = lw.initFullVM(path.encode("utf-8"), True)
            elif option == "client":
                self.vm = lw.initClientVM(path.encode("utf-8"))
            elif option == "server":
                self.vm = lw.initServerVM(path.encode("utf-8"))
            else:
                raise ValueError(option)
        from . import lowlevel

        _live_vms[id(self)] = self.vm
        L = lowlevel.lib()
        self.ctx_handle = lw.hevm_context(self.vm)
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
            lw.load(self.vm, str(const_path).encode("utf-8"), str(hevm_path).encode("utf-8"))
        elif self.option == "client":
            lw.loadClient(self.vm, str(hevm_path).encode("utf-8"))  # the reference passes const_path here (upstream bug)
        if preprocess:
            lw.preprocess(self.vm)
        else:
            raise Exception("Not implemented in SEAL_HEVM")
        self.arglen = lw.getArgLen(self.vm)
        self.reslen = lw.getResLen(self.vm)
        self.hevm_path = str(hevm_path)

    def load_mem(self, cst: bytes, hevm: bytes, preprocess=True):
        """extension: load from memory images (no temp files)"""
        lw.hevm_load_mem(self.vm, cst, len(cst), hevm, len(hevm))
        if preprocess:
            lw.preprocess(self.vm)
        self.arglen = lw.getArgLen(self.vm)
        self.reslen = lw.getResLen(self.vm)
        self.hevm_path = "<memory>"

    def set_streams(self, n):
        """extension: n independent ciphertext streams through the same program (call before load)"""
        lw.hevm_set_streams(self.vm, n)

    def select_stream(self, s):
        lw.hevm_select_stream(self.vm, s)

    def run(self):
        lw.run(self.vm)
        lw.printMem(self.vm)

    def setInput(self, i, data):
        if not isinstance(data, np.ndarray):
            data = np.array(data, dtype=np.float64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        carr = data.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        lw.encrypt(self.vm, i, carr, len(data))

    def setDebug(self, enable):
        lw.setDebug(self.vm, enable)

    def setToGPU(self, ongpu):
        lw.setToGPU(self.vm, ongpu)

    def getOutput(self):
        result = np.zeros((self.reslen, self.slots), dtype=np.float64)  # reference: (reslen, 1 << 14)
        data = np.zeros(self.slots, dtype=np.float64)
        for i in range(self.reslen):
            carr = data.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            lw.decrypt_result(self.vm, i, carr)
            result[i] = data
        return result

    def keyBuffers(self):
        """[(device pointer, 64-bit words)] of the key material in canonical order (hevm_key_buffers)"""
        n = lw.hevm_key_buffers(self.vm, None, None, 0)
        ptrs, words = (ctypes.c_void_p * n)(), (ctypes.c_uint64 * n)()
        lw.hevm_key_buffers(self.vm, ptrs, words, n)
        return [(int(ptrs[i] or 0), int(words[i])) for i in range(n)]

    def keyDigest(self) -> int:
        return int(lw.hevm_key_digest(self.vm))

    def keysReplaced(self):
        lw.hevm_keys_replaced(self.vm)

    def close(self):
        """extension: return this VM's HBM (hevm_destroy).  The reference's runner never frees its VM; neither does this class unless asked."""
        if getattr(self, "vm", None) and _live_vms.get(id(self)) == self.vm:  # (not already released by close_all)
            lw.hevm_destroy(self.vm)
        _live_vms.pop(id(self), None)
        self.vm = None

    def plaintextBytes(self) -> int:
        """extension: HBM held for the program's plaintexts (pre-encoded pool, or constants + window with DACAPO_HEVM_ONLINE_ENCODE=1)"""
        return int(lw.hevm_plaintext_bytes(self.vm))

    def addRotationKeys(self, offsets):
        """extension: direct Galois keys for these slot offsets (create_galois_keys(steps) in SEAL; HEAAN_HEVM.cpp:58-64's key list)"""
        arr = (ctypes.c_int64 * len(offsets))(*[int(o) for o in offsets])
        lw.hevm_add_rotation_keys(self.vm, arr, len(offsets))

    def saveCtxt(self, reg: int, path):
        """extension: seal::Ciphertext::save of a cipher register (SEAL 4.0 bytes) -- what a client / server pair exchanges"""
        lw.hevm_save_ctxt(self.vm, reg, str(path).encode("utf-8"))

    def loadCtxt(self, reg: int, path):
        lw.hevm_load_ctxt(self.vm, reg, str(path).encode("utf-8"))

    def getResIdx(self, i: int) -> int:
        return int(lw.getResIdx(self.vm, i))

    def getCtxt(self, reg: int) -> hevm_ctxt:
        return hevm_ctxt.from_address(lw.getCtxt(self.vm, reg))

    def stats(self):
        counts = (ctypes.c_int64 * 11)()
        ks, ntt = ctypes.c_int64(), ctypes.c_int64()
        lw.hevm_last_run_stats(self.vm, counts, ctypes.byref(ks), ctypes.byref(ntt))
        return {"op_counts": list(counts), "keyswitches": ks.value, "ntts": ntt.value,
                "bootstrap_s": float(lw.hevm_last_run_bootstrap_seconds(self.vm))}

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
