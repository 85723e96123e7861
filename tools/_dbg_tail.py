import os, sys
sys.path.insert(0, "/root/repo")
os.environ["FH_DEBUG_TAIL"] = "1"
import fredholm_amd as F
from fredholm_amd import scenes
cam = F.Camera(**scenes.CORNELL_CAMERA)
w = h = 1024
r = F.Renderer(0); r.load_scene(scenes.cornell_box()); r.build_ias(); r.set_resolution(w, h)
L = F.RenderLayer(r, w, h)
for spp in (8, 8, 1, 1, 1, 8, 8, 1, 1):
    r.reset_stats(); r.render(cam, (0, 0, 0), L, spp, 8); r.wait_for_completion()
    st = r.stats(); print("spp", spp, "tail launches", st["n_tail_launches"], "passes", st["n_passes"], flush=True)
