"""What the predict loop adds to the forwards on a 16-view strip (h16 mode): the strip as bench.py times it, with the PFM writer's
file writes skipped (D2H kept), and with the writer skipped altogether.   python tools/strip_probe.py [casmvsnet adamvs]"""
import os, sys, time, tempfile, shutil
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, predict, synthetic as S

models = sys.argv[1:] or ["casmvsnet", "adamvs"]
ops.set_conv_precision("h16")
strip = predict.SyntheticStrip(16, 5, 2752, 1856, 384, seed=3)
items = [strip[i] for i in range(16)]
tmp = tempfile.mkdtemp(prefix="d3d_probe_")
real_submit, real_run = predict.PfmWriter.submit, predict.PfmWriter._run
def run_nowrite(self):
    while True:
        item = self._q.get()
        if item is None:
            return
        sl, done, paths, display = item
        done.synchronize()
        sl["free"].set()
try:
    for name in models:
        net = predict.build_model(name, 384)
        S.fill_state_dict_(net.state_dict(), 1)
        net = net.cuda().eval()
        predict.predict_views(net, items[:2], os.path.join(tmp, "warm"))
        for mode in ("files", "no file writes", "no writer"):
            predict.PfmWriter.submit = real_submit if mode != "no writer" else (lambda self, maps, paths, display=None: None)
            predict.PfmWriter._run = real_run if mode == "files" else run_nowrite
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            predict.predict_views(net, items, os.path.join(tmp, name + mode.replace(" ", "_")), feature_cache_bytes=32 << 30)
            torch.cuda.synchronize()
            print("%-10s %-15s %.2f ms per view" % (name, mode, (time.perf_counter() - t0) / 16 * 1e3), flush=True)
finally:
    predict.PfmWriter.submit, predict.PfmWriter._run = real_submit, real_run
    shutil.rmtree(tmp, ignore_errors=True)
