#!/usr/bin/env python3
"""Board power and shader clock while the headline kernel (config 2) runs back to back.

Why: a build with twelve compute waves per workgroup needs 18 % fewer shader cycles per workgroup than the production
build (cycle stamps, profiles/r03_ab_variants.txt) and is 5 % SLOWER by wall time -- the chip lowers its clock under the
denser instruction stream (MI355X_MICROARCH.md, 'DVFS give-back').  This probe loops the kernel for SECONDS and samples
`rocm-smi` beside it, so that the clock and the power the chip holds under each build can be read off.

    python tools/power_probe.py [seconds]        (D3D_LIBRARY selects a variant build: tools/run_ab.sh CMD=...)
"""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S  # noqa: E402

V, C, D, H, W = 5, 32, 384, 688, 464


def sample(stop, rows):
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
            j = json.loads(o[o.index("{"):])
            if not rows:
                print("rocm-smi keys:", json.dumps(j)[:600], flush=True)
            card = j[sorted(j)[0]]
            row = {}
            for k, v in card.items():
                kl = k.lower()
                if "power" in kl and "socket" in kl or "average" in kl and "power" in kl:
                    row["power_w"] = float(v)
                if kl.startswith("sclk clock level"):
                    row["sclk"] = v
                if "junction" in kl and "temp" in kl:
                    row["tj"] = float(v)
            rows.append(row)
        except Exception as e:  # noqa: BLE001
            rows.append({"err": str(e)[:80]})
        time.sleep(0.3)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    proj, dv = S.make_scene(V, H, W, D, seed=0)
    feats = [torch.from_numpy(f).cuda() for f in S.make_features(V, C, H, W, seed=0)]
    depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    out = torch.empty((C, D, H, W), dtype=torch.float32, device="cuda")
    for _ in range(5):
        ops.variance_volume(feats, p34, depth, out=out)
    torch.cuda.synchronize()
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, rows))
    th.start()
    t0 = time.perf_counter()
    n = 0
    ms = []
    while time.perf_counter() - t0 < seconds:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.variance_volume(feats, p34, depth, out=out)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b) / 20)
        n += 20
    stop.set()
    th.join()
    pw = [r["power_w"] for r in rows if "power_w" in r]
    print("launches %d  ms/launch first %.3f  median %.3f  last %.3f" % (n, ms[0], float(np.median(ms)), ms[-1]))
    print("power W: n %d  median %.0f  max %.0f | sclk samples: %s | tj: %s" % (
        len(pw), float(np.median(pw)) if pw else -1, max(pw) if pw else -1,
        sorted({str(r.get("sclk")) for r in rows})[:6], sorted({r.get("tj") for r in rows if "tj" in r})[-3:]))
    errs = [r["err"] for r in rows if "err" in r]
    if errs:
        print("sampler errors:", errs[:2])


if __name__ == "__main__":
    main()
