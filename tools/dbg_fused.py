import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import host_scene
from hydracore_amd import HipCore
sc, b = host_scene("atrium_glass_small", 96, 54, 8)
core = HipCore(96, 54, device=0)
core.upload_scene(b)
def render(**opts):
    for k, v in opts.items(): core.set_option(k, v)
    core.set_tile_partition(0, 1, 64); core.init_path_tracing(99); core.reset_perf_counters(); core.trace_pass(3)
    st = core.rays_stat()
    return core.hdr_image(96, 54).copy(), int(st.extensionRays), int(st.shadowRays)
base = render()
for opts in (dict(fused_bounce=0), dict(fused_bounce=1, sort_paths=0), dict(fused_bounce=1, sort_paths=0, scene_tables_in_lds=0), dict(fused_bounce=1, sort_paths=0, scene_tables_in_lds=0, srgb_table=0), dict(fused_bounce=0, srgb_table=0), dict(fused_bounce=0, srgb_table=1, scene_tables_in_lds=1, sort_paths=1)):
    img, e, s = render(**opts)
    d = (img.view(np.uint32) != base[0].view(np.uint32))
    print(opts, "diff px", int(d.any(axis=2).sum()), "rays", (e, s), base[1:], "max abs", float(np.abs(img - base[0]).max()), "nan", int(np.isnan(img).sum()))
    if d.any():
        ys, xs = np.nonzero(d.any(axis=2)); print("  first", ys[:3], xs[:3], img[ys[0], xs[0]], base[0][ys[0], xs[0]])
