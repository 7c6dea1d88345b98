"""The lights' grids by scene: how long their lists are, and the 4096^2 frame with and without them."""
import sys
sys.path.insert(0, ".")
import raytracer_amd as rt
r = rt.Renderer(0)
for scene in ("tetra-3", "tetra-5", "tetra", "teapot-1", "teapot-3", "teapot", "balls", "balls2", "refract", "gears", "ball", "ball2", "2d"):
    try:
        s = rt.Scene.load(scene)
    except Exception as e:
        print(scene, "not loaded:", e); continue
    row = []
    for grid in (1, 3):
        r.set_option("grid", grid)
        r.set_option("verbose", 1 if grid == 1 else 0)
        r.upload(s)
        r.set_option("verbose", 0)
        cam = s.camera(4096, 4096)
        r.render(cam, want=("rgb_u8",))
        row.append(min(r.render(cam, want=("rgb_u8",))["stats"]["gpu_ms"] for _ in range(3)))
    print("%-10s with the lights' grids %.2f ms, without %.2f ms" % (scene, row[0], row[1]), flush=True)
    s.close()
