"""First-frame cost of a scene (upload, hierarchy build, light cameras) against a steady-state frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "glass"
t0 = time.perf_counter(); s = rt.Scene.load(name); t1 = time.perf_counter()
g = rt.Renderer(0); t2 = time.perf_counter()
g.upload(s); t3 = time.perf_counter()
cam = s.camera()
f = g.render(cam, want=("rgb_u8",)); t4 = time.perf_counter()
f = g.render(cam, want=("rgb_u8",)); t5 = time.perf_counter()
print("%s: parse+octree %.1f ms, context %.1f ms, upload+hierarchies %.1f ms, first frame %.1f ms, next frame %.1f ms (light-camera rays %d)"
      % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, f["stats"]["lightcam_rays"]))
