"""One QR call (target of rocprofv3): python tools/qr_one.py M N [dtype]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
be = ta.get_hip_backend()
m, n = int(sys.argv[1]), int(sys.argv[2])
dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
x = be.device_random((m, n), dtype=dt, seed=1)
be.qr(x, 1); be.synchronize()
t0 = time.perf_counter(); be.qr(x, 1); be.synchronize(); print("qr", m, n, time.perf_counter() - t0)
