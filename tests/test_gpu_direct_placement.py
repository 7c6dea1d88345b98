"""The primary pass that takes no queue places from counters (records at places that follow from the chunk's number, the
groups that hold anything listed for the passes that read them) against the pass that reserves, on frames large enough
to be handed out in units: same primary hit ids, ray and hit counts per generation, 8-bit image; the double frame within
1e-9 of 255 (the order of the atomic sums differs).  Scenes: one ray per hit (teapot-3), two (refract, gears: the ray groups
of a chunk are two then), spheres (balls: 128-byte hit records), transparent spheres with light cameras (glass: the shadow
pass decides ray by ray and postpones evaluations); one batch, several batches (a small memory budget), paired passes and
a launch per pass, parts of a frame, super-sampling.

With direct placement the primary pass of a scene whose shadow queries are order-free also forms the lights' terms at its
hits itself (no hit records, no shadow pass for generation 0; option `fuse_lights`): the frames with it (teapot-3, refract and
gears in the form for flat triangles, balls in the general form; glass does not qualify) are compared with the frames without."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def same_frames(a, b):
    assert np.array_equal(a["hit_id"], b["hit_id"])
    assert np.array_equal(a["rgb_u8"], b["rgb_u8"])
    assert np.abs(a["rgb_f64"] - b["rgb_f64"]).max() <= 1e-9
    for k in ("closest_rays", "shadow_rays", "rays_per_generation", "hits_per_generation"):
        assert a["stats"][k] == b["stats"][k], k
    assert a["stats"]["queue_overflow"] == 0 and b["stats"]["queue_overflow"] == 0


@pytest.mark.parametrize("scene,res,spp,pair,budget_gib", [
    ("teapot-3", (2304, 2304), 1, -1, 0), ("teapot-3", (1536, 1536), 2, 0, 0), ("teapot-3", (2304, 2304), 2, -1, 1),
    ("refract", (2304, 2304), 1, -1, 0), ("refract", (2304, 2304), 1, 0, 1), ("gears", (2304, 2304), 1, -1, 0),
    ("balls", (2304, 2304), 1, 0, 0), ("glass", (2304, 2304), 1, -1, 0), ("glass", (2304, 2304), 1, 0, 0)])
def test_direct_placement_gives_the_same_frame(scene, res, spp, pair, budget_gib):
    s = rt.Scene.load(scene)
    g = rt.Renderer(0)
    if budget_gib:
        g.set_memory_budget(budget_gib << 30)
    g.set_option("pair", pair)
    g.upload(s)
    cam = s.camera(*res)
    frames = []
    for direct, fuse in ((0, 1), (1, 0), (1, 1)):
        g.set_option("direct", direct)
        g.set_option("fuse_lights", fuse)
        frames.append(g.render(cam, num_samples=spp, flags=rt.FLAG_KEEP_ZERO_WEIGHT, seed=3))
    same_frames(frames[0], frames[1])
    same_frames(frames[0], frames[2])
    if budget_gib:
        assert frames[2]["stats"]["batches"] > 1
    # a part of the frame (bands that do not line up with the batches)
    parts = [g.render(cam, num_samples=spp, flags=rt.FLAG_KEEP_ZERO_WEIGHT, seed=3, band_rows=24, n_parts=2, part=p) for p in (0, 1)]
    rows = [rt.part_row_indices(res[1], 24, 2, p) for p in (0, 1)]
    for p in (0, 1):
        assert np.array_equal(parts[p]["rgb_u8"], frames[0]["rgb_u8"][rows[p]])
        assert np.array_equal(parts[p]["hit_id"], frames[0]["hit_id"][rows[p]])
    g.close()
    s.close()
