#!/usr/bin/env python3
"""Development timing of the raw kernels through the C ABI (HIP events).  Not the judged bench (bench.py)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from dacapo_amd import lowlevel as ll


def time_ms(fn, iters=20, warm=3):
    L = ll.lib()
    for _ in range(warm):
        fn()
    e0, e1 = L.dc_event_create(), L.dc_event_create()
    L.dc_event_record(e0, None)
    for _ in range(iters):
        fn()
    L.dc_event_record(e1, None)
    return L.dc_event_elapsed_ms(e0, e1) / iters


def main():
    logN = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    ctx = ll.Context(logN, 14)
    N = 1 << logN
    for count in (1, 13, 26, 182, 1024, 4096):
        buf = ll.DeviceBuffer((count, N))
        ll.lib().dc_memset(buf.ptr, 1, buf.nbytes)
        f = time_ms(lambda: ctx.ntt(buf, count, prime_base=0, prime_period=14))
        i = time_ms(lambda: ctx.ntt(buf, count, inverse=True, prime_base=0, prime_period=14))
        gbs = 2 * count * N * 8 / (f * 1e-3) / 1e9
        print(f"N=2^{logN} limbs={count:5d} fwd {f*1e3:9.1f} us ({count/f*1e3:12.0f} NTT/s, {gbs:7.1f} GB/s alg)  inv {i*1e3:9.1f} us")


if __name__ == "__main__":
    main()
