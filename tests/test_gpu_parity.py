"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same scenes.

Bars (per the north-star of this build):
  * primary-ray hit primitive ids: identical (the oracle and the device walk a node's residents
    in the same order, so even exact-t ties resolve the same way);
  * image: >= 99.9 % of pixels within 1/255 per channel -- asserted here in the much stronger
    form "8-bit image identical, f64 frame within 1e-9 of 255", because both sides evaluate the
    same IEEE FP64 operations; only the specular power (within 1.3e-14 of pow(), tests/test_gpu_specular_power.py) and
    the summation order of a pixel's contributions differ;
  * ray counts: equal to the oracle's (with zero-weight reflection rays kept, as the reference does).
"""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

pytestmark = pytest.mark.gpu

SCENES = ["tetra-3", "teapot-3", "balls", "refract"]


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    r.close()


_SCENE_CACHE = {}


def load_scene(gpu, name):
    if name not in _SCENE_CACHE:
        _SCENE_CACHE[name] = rt.Scene.load(name)
    gpu.upload(_SCENE_CACHE[name])
    return _SCENE_CACHE[name]


def assert_frames_match(g, o, tol=1e-9, rows=None):
    o_f64, o_u8, o_hit = o["rgb_f64"], o["rgb_u8"], o["hit_id"]
    if rows is not None:
        o_f64, o_u8, o_hit = o_f64[rows], o_u8[rows], o_hit[rows]
    assert g["rgb_f64"].shape == o_f64.shape
    if "hit_id" in g:
        mism = int((g["hit_id"] != o_hit).sum())
        assert mism == 0, "%d primary hit ids differ" % mism
    diff = np.abs(g["rgb_f64"] - o_f64)
    within = (np.abs(g["rgb_u8"].astype(np.int32) - o_u8.astype(np.int32)).max(axis=2) <= 1).mean()
    assert within >= 0.999, "only %.4f%% of pixels within 1/255" % (100 * within)
    assert diff.max() <= tol, "max |f64 difference| = %g" % diff.max()
    assert np.array_equal(g["rgb_u8"], o_u8)


@pytest.mark.parametrize("scene", SCENES)
def test_scene_matches_oracle(gpu, oracle_frame, scene):
    s = load_scene(gpu, scene)
    g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    o = oracle_frame(scene)
    assert_frames_match(g, o)
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
    assert g["stats"]["queue_overflow"] == 0


# The reference's larger scenes at a reduced resolution (the oracle renders them in seconds):
# gears has 36 k triangles, 65 materials with T = 0.8 and 5 lights (refraction fan-out, a hierarchy
# that does not fit the on-chip node cache); balls2 reflects every ray to the depth limit.
@pytest.mark.parametrize("scene,res", [("teapot", (256, 256)), ("tetra", (300, 300)), ("gears", (160, 160)),
                                       ("balls2", (200, 200))])
def test_large_scenes(gpu, oracle_frame, scene, res):
    s = load_scene(gpu, scene)
    g = gpu.render(s.camera(*res), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    o = oracle_frame(scene, res=res)
    assert_frames_match(g, o)
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
    assert g["stats"]["queue_overflow"] == 0


@pytest.mark.parametrize("scene", ["teapot-3", "balls"])
def test_pruned_walk_equals_reference_walk(gpu, scene):
    """Pruning by entry distance and dropping zero-weight rays change no output value."""
    s = load_scene(gpu, scene)
    a = gpu.render(s.camera(), flags=0)
    b = gpu.render(s.camera(), flags=rt.FLAG_NO_CULL | rt.FLAG_KEEP_ZERO_WEIGHT)
    assert np.array_equal(a["hit_id"], b["hit_id"])
    assert np.array_equal(a["rgb_u8"], b["rgb_u8"])
    assert np.abs(a["rgb_f64"] - b["rgb_f64"]).max() <= 1e-9
    assert a["stats"]["closest_rays"] <= b["stats"]["closest_rays"]


def test_phong_normals(gpu, oracle_frame):
    s = load_scene(gpu, "teapot-3")
    g = gpu.render(s.camera(), flags=rt.FLAG_PHONG_NORMALS | rt.FLAG_KEEP_ZERO_WEIGHT)
    o = oracle_frame("teapot-3", phong=True)
    assert_frames_match(g, o)
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]


@pytest.mark.parametrize("scene", ["glass", "glass2", "glass4"])
def test_transparent_sphere_scenes_without_light_camera(gpu, oracle_frame, scene):
    """Scenes whose shadow rays meet transparent spheres: exact checkOccluded3 ordering.
    The light-camera term itself is switched off on both sides (shadowed instead)."""
    s = load_scene(gpu, scene)
    g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT | rt.FLAG_NO_LIGHTCAM)
    o = oracle_frame(scene, flags=ob.FLAG_NO_LIGHTCAM)
    assert_frames_match(g, o)
    assert np.array_equal(g["lightcam_mask"], o["lightcam_mask"])
    assert g["lightcam_mask"].any()
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]


@pytest.mark.parametrize("scene", ["2d", "teapot-1", "glass3", "ball", "tetra-5", "teapot-2"])
def test_small_scenes(gpu, oracle_frame, scene):
    s = load_scene(gpu, scene)
    g = gpu.render(s.camera(), flags=rt.FLAG_NO_LIGHTCAM)
    o = oracle_frame(scene, flags=ob.FLAG_NO_LIGHTCAM)
    assert_frames_match(g, o)


def test_glass_matches_the_reference_golden_image(gpu, oracle_frame):
    """glass.nff with the light-camera caustics on the device: against the oracle and against
    hide.ppm, the image the reference's authors committed."""
    import gzip
    import os
    s = load_scene(gpu, "glass")
    g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    o = oracle_frame("glass")
    assert_frames_match(g, o, tol=1e-6)
    assert np.array_equal(g["lightcam_mask"], o["lightcam_mask"])
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
    assert g["stats"]["lightcam_rays"] == sum(c["light_rays"] for c in o["lightcams"])
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hide.ppm.gz"), "rb") as f:
        ref = f.read()
    header = b"P6\n900 900\n255\n"
    assert ref.startswith(header)
    golden = np.frombuffer(ref[len(header):], dtype=np.uint8).reshape(900, 900, 3)
    within = (np.abs(g["rgb_u8"].astype(np.int32) - golden.astype(np.int32)).max(axis=2) <= 1).mean()
    assert within >= 0.999
    assert np.array_equal(g["rgb_u8"], golden), "%d bytes differ from hide.ppm" % int((g["rgb_u8"] != golden).sum())


@pytest.mark.parametrize("scene", ["glass2", "glass3"])
def test_light_camera_scenes(gpu, oracle_frame, scene):
    s = load_scene(gpu, scene)
    g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)   # the mask also records zero-weight rays' shading points
    o = oracle_frame(scene)
    assert_frames_match(g, o, tol=1e-6)
    assert np.array_equal(g["lightcam_mask"], o["lightcam_mask"])
    fast = gpu.render(s.camera())
    assert np.array_equal(fast["rgb_u8"], g["rgb_u8"])


def test_density_queries_with_a_full_pool(gpu, oracle_frame):
    """A wave's 64 density queries share one pool of cells still to be visited (LightCam.h:121-203 is a recursion per
    query).  With a tenth of the pool the waves run into its limit all the time: trips then take fewer cells, and the
    frame is the same, with no overflow reported."""
    s = load_scene(gpu, "glass")
    cam = s.camera(300, 300)
    full = gpu.render(cam)
    try:
        gpu.set_option("lightcam_pool", 160)
        tight = gpu.render(cam)
    finally:
        gpu.set_option("lightcam_pool", 1 << 20)
    o = oracle_frame("glass", res=(300, 300))
    assert_frames_match(tight, o, tol=1e-6)
    assert np.array_equal(tight["rgb_u8"], full["rgb_u8"])
    assert tight["stats"]["closest_rays"] == full["stats"]["closest_rays"]


def test_supersampling_and_depth_of_field(gpu, oracle_frame):
    """-s 2 -a 0.3 with the counter-based jitter (the reference's rand() stream is not reproducible)."""
    s = load_scene(gpu, "teapot-3")
    cam = s.camera(128, 128)
    g = gpu.render(cam, num_samples=2, aperture=0.3, seed=7, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    o = oracle_frame("teapot-3", res=(128, 128), num_samples=2, aperture=0.3, seed=7)
    assert_frames_match(g, o, tol=1e-9)
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]


def test_non_square_and_ragged_sizes(gpu, oracle_frame):
    """Frames that are not multiples of the 8x8 launch tile, and resx != resy (integer aspect ratio)."""
    s = load_scene(gpu, "tetra-3")
    for res in [(37, 21), (64, 19), (1, 1), (130, 63)]:
        g = gpu.render(s.camera(*res))
        o = oracle_frame("tetra-3", res=res)
        assert_frames_match(g, o)


def test_row_band_partition_reassembles(gpu, oracle_frame):
    """Row bands dealt round-robin to parts: every part equals the oracle's rows, and together they tile the frame."""
    s = load_scene(gpu, "teapot-3")
    cam = s.camera(96, 100)
    o = oracle_frame("teapot-3", res=(96, 100))
    seen = np.zeros(100, dtype=np.int32)
    for n_parts, band in [(2, 16), (4, 8), (3, 7)]:
        seen[:] = 0
        for part in range(n_parts):
            rows = rt.part_row_indices(100, band, n_parts, part)
            assert rt.part_rows(cam, band, n_parts, part) == len(rows)
            g = gpu.render(cam, band_rows=band, n_parts=n_parts, part=part)
            assert_frames_match(g, o, rows=rows)
            seen[rows] += 1
        assert (seen == 1).all()


@pytest.mark.parametrize("scene", ["teapot-3", "balls", "gears", "refract"])
def test_block_sizes_agree(gpu, scene):
    """Large frames run 1024-thread blocks with one LDS copy of the scene's top per compute unit, small ones
    256-thread blocks: the same frame through both."""
    s = load_scene(gpu, scene)
    cam = s.camera(192, 160)
    frames = []
    for block in (256, 1024):
        gpu.set_option("block_threads", block)
        try:
            frames.append(gpu.render(cam, num_samples=2, flags=rt.FLAG_KEEP_ZERO_WEIGHT))
        finally:
            gpu.set_option("block_threads", 0)
    a, b = frames
    assert np.array_equal(a["hit_id"], b["hit_id"])
    assert np.array_equal(a["rgb_u8"], b["rgb_u8"])
    assert np.abs(a["rgb_f64"] - b["rgb_f64"]).max() <= 1e-9
    assert a["stats"]["closest_rays"] == b["stats"]["closest_rays"] and a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"]


def test_small_batches_equal_one_batch(gpu):
    s = load_scene(gpu, "refract")
    cam = s.camera(160, 160)
    a = gpu.render(cam)
    gpu.set_batch(4096)
    try:
        b = gpu.render(cam)
    finally:
        gpu.set_batch(0)
    assert b["stats"]["batches"] > 1
    assert np.array_equal(a["hit_id"], b["hit_id"])
    assert np.array_equal(a["rgb_u8"], b["rgb_u8"])
    assert a["stats"]["closest_rays"] == b["stats"]["closest_rays"]
    assert a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"]


@pytest.mark.parametrize("scene,res,kw", [
    ("tetra-3", None, {}), ("teapot-3", None, {}), ("balls", (256, 256), {}), ("refract", (256, 256), {}),
    ("gears", (128, 128), {}), ("teapot-3", (100, 90), {"num_samples": 3, "aperture": 0.2, "seed": 5}),
    ("teapot-3", None, {"phong": True})])
def test_one_launch_kernel_equals_oracle(gpu, oracle_frame, scene, res, kw):
    """The kernel that follows a sample's whole ray tree in one launch (chosen by itself only for small frames of scenes
    whose materials neither reflect nor refract) forced on scenes of every kind: same bar as the wavefront pipeline."""
    s = load_scene(gpu, scene)
    phong = kw.pop("phong", False)
    gpu.set_option("fused", 1)
    try:
        cam = s.camera(*res) if res else s.camera()
        g = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT | (rt.FLAG_PHONG_NORMALS if phong else 0), **kw)
        parts = [gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT | (rt.FLAG_PHONG_NORMALS if phong else 0), band_rows=8, n_parts=2,
                            part=p, **kw) for p in range(2)]
    finally:
        gpu.set_option("fused", -1)
    o = oracle_frame(scene, res=res, phong=phong, **kw)
    assert_frames_match(g, o)
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
    for p in range(2):
        assert_frames_match(parts[p], o, rows=rt.part_row_indices(cam.resy, 8, 2, p))


@pytest.mark.parametrize("scene,res", [("teapot-3", None), ("balls", (300, 300)), ("refract", (300, 300)), ("glass3", (256, 256)),
                                       ("gears", (160, 160))])
def test_paired_and_separate_passes_agree(gpu, oracle_frame, scene, res):
    """Small batches run the shadow pass of a generation and the closest-hit pass of the next in one launch, large
    ones a launch per pass (on one stream or two): the same frame through all three, and against the oracle."""
    s = load_scene(gpu, scene)
    cam = s.camera(*res) if res else s.camera()
    frames = []
    for pair, overlap in ((1, -1), (0, 0), (0, 1)):
        gpu.set_option("pair", pair)
        gpu.set_option("overlap", overlap)
        gpu.set_option("fused", 0)
        try:
            frames.append(gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT))
        finally:
            gpu.set_option("pair", -1)
            gpu.set_option("overlap", -1)
            gpu.set_option("fused", -1)
    o = oracle_frame(scene, res=res)
    for g in frames:
        assert_frames_match(g, o, tol=1e-6)
        assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
        assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
        assert np.array_equal(g["lightcam_mask"], o["lightcam_mask"])


@pytest.mark.parametrize("scene,res", [("teapot-3", None), ("balls", (256, 256)), ("refract", (320, 200)), ("gears", (192, 192)),
                                       ("teapot-3", (1200, 1000))])
def test_self_contained_blocks_agree_with_the_passes(gpu, oracle_frame, scene, res):
    """Small frames of scenes whose shadow queries take the hierarchies run as one launch of blocks that each carry
    their own samples through all generations (queues and counters of their own); the launch-per-pass pipeline renders
    the same frame with the same ray counts, and both match the oracle.  The last case has several runs of chunks per
    block."""
    s = load_scene(gpu, scene)
    cam = s.camera(*res) if res else s.camera()
    frames = []
    for local in (1, 0):
        gpu.set_option("local", local)
        gpu.set_option("fused", 0)
        try:
            frames.append(gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT))
        finally:
            gpu.set_option("local", -1)
            gpu.set_option("fused", -1)
    o = oracle_frame(scene, res=res)
    for g in frames:
        assert_frames_match(g, o, tol=1e-6)
        assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
        assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
        assert g["stats"]["queue_overflow"] == 0
    assert frames[0]["stats"]["rays_per_generation"] == frames[1]["stats"]["rays_per_generation"]
    assert frames[0]["stats"]["hits_per_generation"] == frames[1]["stats"]["hits_per_generation"]


@pytest.mark.parametrize("scene,res,n", [("gears", (192, 160), 2), ("refract", (256, 256), 1), ("balls", (200, 200), 1)])
def test_self_contained_blocks_are_repeatable(gpu, scene, res, n):
    """The waves of a self-contained block hand work to each other through counters in LDS, with no barrier; a chunk taken
    before what it reads is written, or a list overwritten while it is still read, would show as a run whose ray counts
    or pixels differ.  Every block size, many runs, against the launch-per-pass frame."""
    s = load_scene(gpu, scene)
    cam = s.camera(*res)
    gpu.set_option("fused", 0)
    try:
        gpu.set_option("local", 0)
        ref = gpu.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        gpu.set_option("local", 1)
        for threads in (1024, 512, 256):
            gpu.set_option("local_threads", threads)
            for _ in range(12):
                f = gpu.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
                assert f["stats"]["rays_per_generation"] == ref["stats"]["rays_per_generation"], threads
                assert f["stats"]["shadow_rays"] == ref["stats"]["shadow_rays"], threads
                assert np.array_equal(f["rgb_u8"], ref["rgb_u8"]), threads
                assert np.abs(f["rgb_f64"] - ref["rgb_f64"]).max() <= 1e-9, threads
    finally:
        gpu.set_option("local", -1)
        gpu.set_option("local_threads", 1024)
        gpu.set_option("fused", -1)


@pytest.mark.parametrize("scene,res", [("balls", (200, 136)), ("teapot-3", (333, 250)), ("refract", (640, 480))])
def test_self_contained_blocks_share_out_any_frame(gpu, scene, res):
    """The plan of the self-contained blocks (how many blocks, how far a block's share of the primary chunks moves on from
    round to round) decides who renders what, never what is rendered: frames whose chunk count is no multiple of the grid,
    with every rotation and minimum share, against the launch-per-pass frame."""
    s = load_scene(gpu, scene)
    cam = s.camera(*res)
    gpu.set_option("fused", 0)
    try:
        gpu.set_option("local", 0)
        ref = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        gpu.set_option("local", 1)
        for rotate, min_chunks in ((-1, 0), (0, 0), (1, 1), (7, 3), (255, 16), (1000, 64), (31, 5)):
            gpu.set_option("local_rotate", rotate)
            gpu.set_option("local_min_chunks", min_chunks)
            f = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
            assert f["stats"]["queue_overflow"] == 0
            assert f["stats"]["rays_per_generation"] == ref["stats"]["rays_per_generation"], (rotate, min_chunks)
            assert f["stats"]["shadow_rays"] == ref["stats"]["shadow_rays"], (rotate, min_chunks)
            assert np.array_equal(f["hit_id"], ref["hit_id"]), (rotate, min_chunks)
            assert np.array_equal(f["rgb_u8"], ref["rgb_u8"]), (rotate, min_chunks)
            assert np.abs(f["rgb_f64"] - ref["rgb_f64"]).max() <= 1e-9, (rotate, min_chunks)
    finally:
        gpu.set_option("local", -1)
        gpu.set_option("local_rotate", -1)
        gpu.set_option("local_min_chunks", 0)
        gpu.set_option("fused", -1)


@pytest.mark.parametrize("scene,res,n", [("teapot-3", (300, 200), 2), ("refract", (256, 192), 1), ("balls", (200, 136), 1), ("glass", (160, 160), 1)])
def test_primary_pass_pushes_whole_chunks(gpu, oracle_frame, scene, res, n):
    """The closest-hit passes of the first generations round what a wave pushes up to whole chunks of 64 records and mark the
    fillers, so that a chunk of the passes that follow is one wave's output; the fillers are no rays and no hits: same frame, same ray and hit counts
    as with packed records and as the oracle, on every path (self-contained blocks, paired passes, a launch per pass)."""
    s = load_scene(gpu, scene)
    cam = s.camera(*res)
    o = oracle_frame(scene, res=res, num_samples=n)
    gpu.set_option("fused", 0)
    try:
        for path in ({"local": 1}, {"local": 0, "pair": 1}, {"local": 0, "pair": 0}):
            for k, v in path.items():
                gpu.set_option(k, v)
            frames = []
            for whole in (-1, 6, 1, 0):
                gpu.set_option("whole_chunks", whole)
                frames.append(gpu.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT))
            for g in frames:
                assert_frames_match(g, o, tol=1e-6)
                assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"], path
                assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"], path
                assert g["stats"]["queue_overflow"] == 0
            for f in frames[:-1]:
                assert f["stats"]["rays_per_generation"] == frames[-1]["stats"]["rays_per_generation"], path
                assert f["stats"]["hits_per_generation"] == frames[-1]["stats"]["hits_per_generation"], path
                assert np.array_equal(f["rgb_u8"], frames[-1]["rgb_u8"]), path
    finally:
        gpu.set_option("whole_chunks", -1)
        gpu.set_option("local", -1)
        gpu.set_option("pair", -1)
        gpu.set_option("fused", -1)


def test_primary_pass_in_units_drawn_while_it_runs(gpu, oracle_frame):
    """A primary pass with many chunks per wave hands its chunks out in units while it runs (a counter in memory, a
    permuted order, records sharded by unit) instead of giving every block a fixed share: the same frame and ray counts
    both ways, and against the oracle."""
    scene, res, n = "teapot-3", (2048, 1040), 2          # 8.5 M samples: above the threshold for units
    s = load_scene(gpu, scene)
    cam = s.camera(*res)
    frames = []
    for units in (1, 0):
        gpu.set_option("units", units)
        try:
            frames.append(gpu.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT))
        finally:
            gpu.set_option("units", 1)
    o = oracle_frame(scene, res=res, num_samples=n)
    for g in frames:
        assert_frames_match(g, o, tol=1e-6)
        assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
        assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
    assert np.array_equal(frames[0]["rgb_u8"], frames[1]["rgb_u8"])
    assert frames[0]["stats"]["rays_per_generation"] == frames[1]["stats"]["rays_per_generation"]


def test_stale_ring_words_are_not_taken_for_the_awaited_unit(gpu, oracle_frame):
    """The hand-over of units inside a block goes through four LDS words tagged with the unit's place in the block's
    sequence.  LDS keeps what the last block on a compute unit left there; with the option `ring_poison` a launch before
    every such primary pass fills those words, on every compute unit, with exactly the tags the pass's blocks will wait
    for and the number of a unit that is not theirs.  A pass that took such a word for the awaited one would trace that
    unit instead of its own: wrong hit map, wrong ray counts, on the first frame."""
    scene, res, n = "teapot-3", (2048, 1040), 2
    s = load_scene(gpu, scene)
    cam = s.camera(*res)
    gpu.set_option("ring_poison", 1)
    try:
        g = gpu.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    finally:
        gpu.set_option("ring_poison", 0)
    o = oracle_frame(scene, res=res, num_samples=n)
    assert g["stats"]["queue_overflow"] == 0
    assert_frames_match(g, o, tol=1e-6)
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]


def test_host_built_light_camera_trees_give_the_same_frame(gpu, oracle_frame):
    """The light cameras' point octrees are built on the device; the host builder with the same rules (Octree.h:1394-1497)
    stays behind the option `lightcam_host`.  glass.nff with it: the reference's own hide.ppm again."""
    import gzip
    import os
    s = load_scene(gpu, "glass")
    gpu.set_option("lightcam_host", 1)       # (drops the cameras the context holds; they are set up again on demand)
    try:
        g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    finally:
        gpu.set_option("lightcam_host", 0)
    o = oracle_frame("glass")
    assert_frames_match(g, o, tol=1e-6)
    assert g["stats"]["lightcam_rays"] == sum(c["light_rays"] for c in o["lightcams"])
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hide.ppm.gz"), "rb") as f:
        ref = f.read()
    golden = np.frombuffer(ref[len(b"P6\n900 900\n255\n"):], dtype=np.uint8).reshape(900, 900, 3)
    assert np.array_equal(g["rgb_u8"], golden)


@pytest.mark.parametrize("option,value", [("lightcam_levels", 1), ("lightcam_digits", 9)])
def test_level_by_level_light_camera_trees_give_the_same_frame(gpu, oracle_frame, option, value):
    """The device builds the light cameras' point octrees with two sorts of the points' whole ways down the tree; the builder
    that sorts once per level stays behind `lightcam_levels`, and takes over by itself when a tree is deeper than the keys
    of the first can say (`lightcam_digits` lowers that depth to one glass.nff's trees exceed).  Either way: hide.ppm."""
    import gzip
    import os
    s = load_scene(gpu, "glass")
    gpu.set_option(option, value)
    try:
        g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    finally:
        gpu.set_option(option, 0)
    o = oracle_frame("glass")
    assert_frames_match(g, o, tol=1e-6)
    assert g["stats"]["lightcam_rays"] == sum(c["light_rays"] for c in o["lightcams"])
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hide.ppm.gz"), "rb") as f:
        ref = f.read()
    golden = np.frombuffer(ref[len(b"P6\n900 900\n255\n"):], dtype=np.uint8).reshape(900, 900, 3)
    assert np.array_equal(g["rgb_u8"], golden)


@pytest.mark.parametrize("scene", ["glass3", "glass"])
def test_one_launch_kernel_with_light_cameras(gpu, oracle_frame, scene):
    s = load_scene(gpu, scene)
    gpu.set_option("fused", 1)
    try:
        g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    finally:
        gpu.set_option("fused", -1)
    o = oracle_frame(scene)
    assert_frames_match(g, o, tol=1e-6)
    assert np.array_equal(g["lightcam_mask"], o["lightcam_mask"])
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]


def test_queue_overflow_is_detected_and_rendered_again(gpu, oracle_frame):
    """Ray queues are sized for a multiple of the primary count per generation, not for the 32-fold worst case.
    With the multiple forced to 1 the refraction rays of refract.nff outgrow them: the asynchronous path reports
    RT_ERR_CAPACITY (and deepens the queues for the next frame), rt_render repeats the frame by itself."""
    import torch
    s = load_scene(gpu, "refract")
    cam = s.camera(256, 256)
    o = oracle_frame("refract", res=(256, 256))
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    gpu.set_option("queue_growth", 1)
    with pytest.raises(rt.RtError) as e:
        gpu.render_device(cam, rgb_u8_ptr=u8.data_ptr(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    assert e.value.code == -4
    st = gpu.render_device(cam, rgb_u8_ptr=u8.data_ptr(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)     # depth 2 now: complete
    assert st["queue_overflow"] == 0
    assert np.array_equal(u8.cpu().numpy(), o["rgb_u8"])
    gpu.set_option("queue_growth", 1)
    g = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)           # overflows once, is rendered again inside the call
    assert_frames_match(g, o)
    assert g["stats"]["queue_overflow"] == 0
    assert g["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    load_scene(gpu, "refract")                                    # an upload restores the default depth


def test_memory_budget_only_changes_the_batching(gpu):
    s = load_scene(gpu, "balls")
    cam = s.camera(320, 320)
    a = gpu.render(cam)
    gpu.set_memory_budget(24 << 20)
    try:
        b = gpu.render(cam)
    finally:
        gpu.set_memory_budget(0)
    assert b["stats"]["batches"] > a["stats"]["batches"]
    assert np.array_equal(a["hit_id"], b["hit_id"])
    assert np.array_equal(a["rgb_u8"], b["rgb_u8"])
    assert a["stats"]["closest_rays"] == b["stats"]["closest_rays"] and a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"]


def test_light_cameras_are_set_up_on_demand(gpu, oracle_mod):
    """4200 (transparent sphere, light) pairs, of which a frame asks for a fraction: like the reference (Camera.h:417-440)
    the device sets up the cameras a frame's shading points need, when they need them, and the frame equals the
    oracle's; a second view adds the cameras it needs to the ones that exist."""
    import os
    import tempfile
    lines = ["b 0.1 0.1 0.2", "v", "from 0 -12 4", "at 0 0 0", "up 0 0 1", "angle 40", "hither 1", "resolution 56 56", "Z 24"]
    for k in range(5):
        lines.append("l %g 6 9" % (k - 2.0))
    lines.append("f 0.8 0.8 0.8 0.9 0.1 10 0 1")
    lines += ["p 4", "-8 -8 -1", "8 -8 -1", "8 8 -1", "-8 8 -1"]
    lines.append("f 0.9 0.9 1 0.1 0.2 50 0.9 1.3")
    n = 0
    for i in range(30):
        for j in range(28):
            lines.append("s %g %g 0 0.1" % (-4 + 0.27 * i, -4 + 0.27 * j))
            n += 1
    assert n * 5 > 4096
    path = os.path.join(tempfile.mkdtemp(), "many_glass.nff")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    s = rt.Scene.load(path)
    gpu.upload(s)
    o = oracle_mod.OracleScene(path)
    ref = o.render(threads=min(16, os.cpu_count() or 1))
    cams = o.lightcams()
    o.close()
    g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    assert g["lightcam_mask"].any()
    assert_frames_match(g, ref, tol=1e-6)
    assert np.array_equal(g["lightcam_mask"], ref["lightcam_mask"])
    assert 0 < len(cams) < n * 5
    assert g["stats"]["lightcam_rays"] == sum(c["light_rays"] for c in cams)     # the same cameras, no more
    # a narrower view of the same scene: cameras already there are kept, new ones may be added, the frame is the oracle's
    o = oracle_mod.OracleScene(path)
    o.set_resolution(40, 72)
    ref2 = o.render(threads=min(16, os.cpu_count() or 1))
    o.close()
    g2 = gpu.render(s.camera(40, 72), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    assert_frames_match(g2, ref2, tol=1e-6)
    assert g2["stats"]["lightcam_rays"] >= g["stats"]["lightcam_rays"]
    s.close()


def test_bad_arguments_are_errors(gpu):
    s = load_scene(gpu, "tetra-3")
    cam = s.camera()
    with pytest.raises(rt.RtError):
        gpu.render(cam, num_samples=0)
    with pytest.raises(rt.RtError):
        gpu.render(cam, n_parts=2, part=2)
    with pytest.raises(rt.RtError):
        rt.Renderer(99)
    fresh = rt.Renderer(0)
    with pytest.raises(rt.RtError):
        fresh.render(cam)   # no scene uploaded
    fresh.close()


def test_async_render_into_device_buffers_and_two_contexts(gpu, oracle_frame):
    """rt_render_async + rt_sync with torch device buffers; a second context on the same GPU renders
    another scene in between and neither disturbs the other."""
    import torch
    s = load_scene(gpu, "teapot-3")
    cam = s.camera()
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    f64 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.float64, device="cuda")
    hit = torch.full((cam.resy, cam.resx), -2, dtype=torch.int32, device="cuda")
    assert gpu.render_device(cam, rgb_f64_ptr=f64.data_ptr(), rgb_u8_ptr=u8.data_ptr(), hit_id_ptr=hit.data_ptr(),
                             flags=rt.FLAG_KEEP_ZERO_WEIGHT, sync=False) is None
    with pytest.raises(rt.RtError):            # a frame is in flight: a second one is refused, not queued
        gpu.render_device(cam, rgb_u8_ptr=u8.data_ptr(), sync=False)
    other = rt.Renderer(0)
    s2 = rt.Scene.load("balls")
    other.upload(s2)
    g2 = other.render(s2.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    st = gpu.sync()
    with pytest.raises(rt.RtError):
        gpu.sync()                              # nothing in flight any more
    o = oracle_frame("teapot-3")
    assert np.array_equal(u8.cpu().numpy(), o["rgb_u8"])
    assert np.array_equal(hit.cpu().numpy(), o["hit_id"])
    assert np.abs(f64.cpu().numpy() - o["rgb_f64"]).max() <= 1e-9
    assert st["closest_rays"] == o["counters"]["closest_rays"]
    assert np.array_equal(g2["rgb_u8"], oracle_frame("balls")["rgb_u8"])
    other.close()
    s2.close()


def test_context_reuse_across_scenes_sizes_and_modes(gpu, oracle_frame):
    """One context through a sequence a long-running host would produce: scenes replaced, frame sizes going up and
    down, the normal-interpolation flag toggled on a scene with light cameras (they see it and are rebuilt), a frame
    left in flight when its context is destroyed.  Every frame must equal the one a fresh context renders."""
    import torch
    ctx = rt.Renderer(0)
    with pytest.raises(rt.RtError):
        ctx.render(rt.Scene.load("tetra-3").camera())            # nothing uploaded yet
    seq = [("balls", (96, 96), 0), ("glass3", (120, 100), 0), ("glass3", (60, 50), rt.FLAG_PHONG_NORMALS), ("glass3", (120, 100), 0),
           ("teapot-3", (200, 160), rt.FLAG_PHONG_NORMALS), ("tetra-3", (64, 64), 0), ("balls", (96, 96), 0)]
    current = None
    for name, res, flags in seq:
        s = _SCENE_CACHE.setdefault(name, rt.Scene.load(name))
        if name != current:
            ctx.upload(s)
            current = name
        a = ctx.render(s.camera(*res), flags=flags)
        fresh = rt.Renderer(0)
        fresh.upload(s)
        b = fresh.render(s.camera(*res), flags=flags)
        fresh.close()
        assert np.array_equal(a["hit_id"], b["hit_id"]), (name, res)
        assert np.array_equal(a["rgb_u8"], b["rgb_u8"]), (name, res)
        assert np.abs(a["rgb_f64"] - b["rgb_f64"]).max() <= 1e-9, (name, res)   # the order of the atomic sums is free
        assert np.array_equal(a["lightcam_mask"], b["lightcam_mask"]), (name, res)
    o = oracle_frame("balls", res=(96, 96))
    assert np.array_equal(a["rgb_u8"], o["rgb_u8"])
    # a frame in flight when the context goes away: the library waits for it, nothing is written afterwards
    s = _SCENE_CACHE["balls"]
    cam = s.camera(512, 512)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    ctx.render_device(cam, rgb_u8_ptr=u8.data_ptr(), sync=False)
    ctx.close()
    torch.cuda.synchronize()
    assert np.array_equal(u8.cpu().numpy(), oracle_frame("balls")["rgb_u8"])
