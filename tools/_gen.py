import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amrvolumerenderer_amd import runtime, scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.renderer import build_scene_on_device
from amrvolumerenderer_amd.types import make_params
for cfg, size, tr in (("config4", 2048, 0.97), ("config4", 2048, 0.0), ("config3", 2048, 0.97), ("config2", 1024, 0.97)):
    spec = getattr(scenes, cfg)("smooth")
    scenes.assign_owners(spec, 1, "level_pairs")
    ctx = runtime.Context(0)
    all_boxes, local = build_scene_on_device(ctx, spec, 0)
    ref = runtime.reference_sample_distance(all_boxes, spec.bounds.min_corner, spec.bounds.max_corner)
    params = make_params(size, size, spec.scalar_range, tr, ref, spec.bounds)
    plan = FramePlan(all_boxes, params, scenes.default_camera(), 0, 1)
    scene = ctx.create_scene(local, spec.transform)
    counters = torch.zeros(5, dtype=torch.int64, device=ctx.device)
    samples = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    ctx.set_march_counters(counters)
    scene.render_plan(plan, samples=samples)
    ctx.synchronize()
    ctx.set_march_counters(None)
    total = int(samples.item()); general = int(counters[0].item())
    print(cfg, tr, "samples", total, "in the general loop", general, f"{100.0 * general / total:.2f} %")
    del scene, local, all_boxes
    torch.cuda.empty_cache()
