"""Prints traversal statistics of one frame from a -DRT_COUNT build of the library.
usage: RAYTRACER_AMD_LIB=build_variants/lib_count.so python tools/walk_stats.py [workload]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
import bench
args = [a for a in sys.argv[1:] if not a.startswith("--")]
w = args[0] if args else "teapot3"
json_out = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")]
scene_name, resx, resy, N, _, _ = bench.WORKLOADS[w]
lib = rt.load()
fn = C.CDLL(rt.LIB_PATH).rt_debug_counters
fn.argtypes = [C.POINTER(C.c_ulonglong)]
s = rt.Scene.load(scene_name); g = rt.Renderer(0); g.upload(s); g.set_option("fused", 0)
cam = s.camera(resx, resy)
buf = (C.c_ulonglong * 48)()
g.render(cam, num_samples=N, want=("rgb_u8",))      # (a small frame reads the eye's grid from its camera's second frame on)
fn(buf)
fr = g.render(cam, num_samples=N, want=("rgb_u8",))
fn(buf)
names = ["rays", "lane_nodes", "wave_nodes", "lane_prims", "wave_prims", "lane_leaves", "wave_rounds"]
print(w, fr["stats"]["closest_rays"], fr["stats"]["shadow_rays"])
record = {"_meta": {"build_id": rt.build_id(), "workload": w}, "lights": int(s.desc.n_lights)}
tri_share = 1.0 if s.desc.n_sph_slots == 0 else (0.0 if s.desc.n_tri_slots == 0 else 0.5)
for gi, gname in enumerate(("primary", "queued", "shadow")):
    v = dict(zip(names, list(buf[gi * 8:gi * 8 + 7])))
    r = max(v["rays"], 1)
    record[gname] = {"rays": v["rays"], "nodes_per_ray": v["lane_nodes"] / r, "leaves_per_ray": v["lane_leaves"] / r,
                     "tri_tests_per_ray": v["lane_prims"] / r * tri_share, "sph_tests_per_ray": v["lane_prims"] / r * (1 - tri_share),
                     "wave_node_steps_per_64_rays": v["wave_nodes"] * 64 / r, "wave_leaf_tests_per_64_rays": v["wave_prims"] * 64 / r,
                     "wave_rounds_per_64_rays": v["wave_rounds"] * 64 / r}
    print("%-8s rays %11d | per ray: nodes %.2f leaves %.2f prims %.2f | per 64 rays the wave ran: node steps %.1f prim tests %.1f rounds %.1f"
          % (gname, v["rays"], v["lane_nodes"] / r, v["lane_leaves"] / r, v["lane_prims"] / r,
             v["wave_nodes"] * 64 / r, v["wave_prims"] * 64 / r, v["wave_rounds"] * 64 / r))
q = list(buf[24:32])
if q[0]:
    n = q[0]
    print("lightcam queries %d | per query: descent nodes %.1f, cells inside reached nodes %.1f, whole-cell sums %.1f, "
          "points tested %.1f, ranges scanned by the wave: %d in all" % (n, q[1] / n, q[2] / n, q[3] / n, q[4] / n, q[5]))
t = list(buf[32:40])
if t[0]:
    w = t[0]
    print("lightcam waves %d | per wave: %.0f ticks (100 MHz clock) in its 64 queries, %.0f of them in %.1f cooperative phases, %.1f trips "
          "with %.1f of 64 lanes holding a cell | longest wave %d ticks, %d trips | most cells pending in a wave %d" % (w, t[1] / w, t[2] / w, t[4] / w, t[3] / w, q[6] / max(t[3], 1), t[5], t[6], q[7]))
if t[0] and buf[42] + buf[44]:
    print("lightcam trips: %.1f per wave judge children (%.0f ticks each), %.1f per wave sum cells (%.0f ticks each)"
          % (buf[42] / t[0], buf[41] / max(buf[42], 1), buf[44] / t[0], buf[43] / max(buf[44], 1)))
if json_out:
    import json
    with open(json_out[0], "w") as f:
        json.dump(record, f, indent=1, sort_keys=True)
