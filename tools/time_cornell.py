"""BASELINE configs[1]: Cornell box + area light, 1080p, 256 spp, Standard Surface + MIS NEE -- throughput on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fredholm_amd as F
from fredholm_amd import native as N, scenes
W, H, SPP, DEPTH = 1920, 1080, 256, int(os.environ.get("DEPTH", "8"))
r = F.Renderer(0); r.load_scene(scenes.cornell_box()); r.build_ias()
r.set_resolution(W, H)
r.set_path_pool(W * H * 32)
L = F.RenderLayer(r, W, H)
cam = F.Camera(**scenes.CORNELL_CAMERA)
for _ in range(2):
    r.render(cam, (0, 0, 0), L, SPP, DEPTH)
r.wait_for_completion()
r.set_flags(N.FLAG_TIME_KERNELS); r.reset_stats()
t0 = time.perf_counter(); n = 4
for _ in range(n):
    r.render(cam, (0, 0, 0), L, SPP, DEPTH)
r.wait_for_completion()
dt = (time.perf_counter() - t0) / n
st = r.stats()
print(f"cornell 1080p {SPP} spp depth {DEPTH}: {dt*1e3:.1f} ms per frame, {W*H*SPP/dt/1e6:.1f} Msamples/s", {k: round(v / n, 2) for k, v in st.items() if k.endswith('_ms') and 'bvh' not in k})
b = L.download("beauty"); print("mean", float(np.nanmean(b[..., :3])), "nan", int(np.isnan(b).sum()))
