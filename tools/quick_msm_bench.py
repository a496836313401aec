"""Development timing probe (not the judged bench): MSM time breakdown at a few sizes."""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gosnark_amd
from gosnark_amd import capi
import gpu_util as U

capi.init()
for g2 in (False, True):
    for logn in ([12, 16, 18, 20] if not g2 else [12, 16, 18]):
        n = 1 << logn
        k = U.rand_scalars_u64(n, 1)
        t0 = time.time()
        bases = (capi.g2_fixed_base if g2 else capi.g1_fixed_base)(k)
        tfb = time.time() - t0
        s = capi.scalars_upload(U.rand_scalars_u64(n, 2))
        for cbits in ([0] if logn < 20 else [0, 16, 15]):
            capi.set_window_bits(cbits)
            capi.msm_resident(bases, s, n, g2=g2)
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                capi.msm_resident(bases, s, n, g2=g2)
            wall = (time.time() - t0) / reps * 1e3
            tm = capi.last_timing()
            print("G%d n=2^%d c=%d fixed_base %.1f ms | msm wall %.2f ms device total %.2f plan %.2f acc %.2f red %.2f | %.2f Mterm/s"
                  % (2 if g2 else 1, logn, cbits, tfb * 1e3, wall, tm["total_ms"], tm["plan_ms"], tm["accumulate_ms"], tm["reduce_ms"], n / wall / 1e3), flush=True)
        capi.set_window_bits(0)
