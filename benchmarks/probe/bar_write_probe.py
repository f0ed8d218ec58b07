"""Can the HOST store straight into device memory (large-BAR mapping) on this box? One subprocess per allocation kind: a host
memset into the device pointer, read back with hipMemcpy. A segmentation fault = not mapped for the CPU."""
import ctypes as C
import subprocess
import sys
import time

KINDS = {"hipMalloc": None, "finegrained": 0x1, "uncached": 0x3, "managed": "managed"}

if len(sys.argv) > 1:
    kind = sys.argv[1]
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    if KINDS[kind] is None:
        rc = hip.hipMalloc(C.byref(p), 4096)
    elif KINDS[kind] == "managed":
        rc = hip.hipMallocManaged(C.byref(p), 4096, 1)
    else:
        rc = hip.hipExtMallocWithFlags(C.byref(p), 4096, KINDS[kind])
    print(kind, "alloc rc", rc, hex(p.value or 0), flush=True)
    hip.hipMemset(p, 0, 4096)
    hip.hipDeviceSynchronize()
    C.memset(p.value, 0x5A, 256)                      # the host store
    t0 = time.perf_counter()
    for _ in range(1000):
        C.memset(p.value, 0x5B, 256)
    dt = (time.perf_counter() - t0) / 1000
    buf = (C.c_ubyte * 256)()
    rc = hip.hipMemcpy(buf, p, 256, 2)
    print(kind, "host store ok; device sees", hex(buf[0]), hex(buf[255]), "memcpy rc", rc, f"host write of 256 B: {dt * 1e6:.2f} us (incl. ctypes)", flush=True)
    sys.exit(0)

for k in KINDS:
    r = subprocess.run([sys.executable, __file__, k], capture_output=True, text=True)
    print(k, "-> rc", r.returncode, "|", r.stdout.strip().replace("\n", " | "), "|", r.stderr.strip()[-200:])
