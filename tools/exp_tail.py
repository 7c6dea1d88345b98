"""How full the resident wave slots are over the shadow launches of the 8K x 16 spp frame (-DRT_COUNT build).
usage: RAYTRACER_AMD_LIB=build_variants/lib_count.so python tools/exp_tail.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
lib = rt.load()
fn = C.CDLL(rt.LIB_PATH).rt_debug_counters
fn.argtypes = [C.POINTER(C.c_ulonglong)]
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
buf = (C.c_ulonglong * 48)()
g.render(cam, num_samples=4, want=("rgb_u8",))
fn(buf)
st = g.render(cam, num_samples=4, want=("rgb_u8",), flags=rt.FLAG_TIMING)["stats"]
fn(buf)
busy_s = buf[40] / 100e6
waves_resident = 256 * 16
print("shadow: sum of wave lifetimes %.3f s, launches %.2f ms x %d resident waves = %.3f s -> %.1f %% of the wave slots busy"
      % (busy_s, st["shadow_ms"], waves_resident, st["shadow_ms"] * 1e-3 * waves_resident, 100 * busy_s / (st["shadow_ms"] * 1e-3 * waves_resident)))
