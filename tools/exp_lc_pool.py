"""Light-camera density pass of several scenes with a -DRT_COUNT build: trips per wave, lanes filled, most cells pending.
usage: RAYTRACER_AMD_LIB=build_variants/lib_count.so python tools/exp_lc_pool.py [scene[:resx:resy] ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
lib = rt.load()
fn = C.CDLL(rt.LIB_PATH).rt_debug_counters
fn.argtypes = [C.POINTER(C.c_ulonglong)]
for spec in sys.argv[1:] or ["glass", "glass2", "glass4"]:
    p = spec.split(":")
    s = rt.Scene.load(p[0]); g = rt.Renderer(0); g.upload(s)
    cam = s.camera(int(p[1]), int(p[2])) if len(p) == 3 else s.camera(0, 0)
    buf = (C.c_ulonglong * 48)()
    fn(buf)        # reading clears
    fr = g.render(cam, num_samples=1, want=("rgb_u8",))
    fn(buf)
    q, t = list(buf[24:32]), list(buf[32:40])
    w = max(t[0], 1)
    print("%-16s queries %8d waves %6d | trips per wave %.1f (most %d), lanes filled %.1f | most cells pending %d of the pool | error %s"
          % (spec, q[0], t[0], t[3] / w, t[6], q[6] / max(t[3], 1), q[7], fr["stats"].get("error", "-")))
