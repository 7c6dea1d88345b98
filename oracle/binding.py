"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py, as the checker.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "librt_oracle.so")
FLAG_NO_LIGHTCAM = 1

_lib = None


def build():
    subprocess.run(["make", "-C", _DIR], check=True, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.orc_load.restype = C.c_void_p
    lib.orc_load.argtypes = [C.c_char_p, C.c_int]
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_info.argtypes = [C.c_void_p, ip]
    lib.orc_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_set_lightcam_samples.argtypes = [C.c_void_p, C.c_int]
    lib.orc_render.restype = C.c_int
    lib.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int,
                               dp, ip, dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_uint64)]
    lib.orc_render_rows.restype = C.c_int
    lib.orc_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, ip, C.c_int, C.c_uint32, C.c_int,
                                    dp, ip, dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_uint64)]
    lib.orc_write_ppm.argtypes = [dp, C.c_int, C.c_int, C.c_char_p]
    lib.orc_get_solid.argtypes = [C.c_void_p, C.c_int, ip, dp, dp, ip, dp]
    lib.orc_get_camera.argtypes = [C.c_void_p, dp]
    lib.orc_get_light.argtypes = [C.c_void_p, C.c_int, dp]
    lib.orc_get_background.argtypes = [C.c_void_p, dp]
    lib.orc_dump_octree.restype = C.c_int
    lib.orc_dump_octree.argtypes = [C.c_void_p, dp, ip, ip]
    lib.orc_test_triangle.restype = C.c_double
    lib.orc_test_triangle.argtypes = [dp, dp, C.c_int, dp, dp, C.c_double, C.c_double, dp]
    lib.orc_test_sphere.restype = C.c_double
    lib.orc_test_sphere.argtypes = [dp, C.c_double, dp, dp, C.c_double, C.c_double, dp]
    lib.orc_test_bound.restype = C.c_double
    lib.orc_test_bound.argtypes = [dp, C.c_double, dp, dp, C.c_double, C.c_double]
    lib.orc_test_batch.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, dp]
    lib.orc_lightcam_count.restype = C.c_int
    lib.orc_lightcam_count.argtypes = [C.c_void_p]
    lib.orc_lightcam_info.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, C.POINTER(C.c_uint64)]
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


COUNTER_NAMES = ("closest_rays", "shadow_rays", "node_tests_closest", "prim_tests_closest",
                 "node_tests_shadow", "prim_tests_shadow", "lightcam_queries")


class OracleScene:
    def __init__(self, path, phong=False):
        self._h = load().orc_load(path.encode(), 1 if phong else 0)
        if not self._h:
            raise IOError("oracle: cannot open " + path)
        info = (C.c_int * 8)()
        load().orc_info(self._h, info)
        (self.resx, self.resy, self.n_solids, self.n_nodes, self.n_lights, self.lightcam_samples,
         self.n_triangles, self.n_spheres) = list(info)

    def close(self):
        if self._h:
            load().orc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_resolution(self, resx, resy):
        load().orc_set_resolution(self._h, resx, resy)
        self.resx, self.resy = resx, resy

    def set_lightcam_samples(self, n):
        load().orc_set_lightcam_samples(self._h, n)
        self.lightcam_samples = n

    def render(self, num_samples=1, aperture=0.0, threads=8, row0=0, row1=None, seed=0, flags=0,
               want_hits=True, want_mask=True, want_u8=True, rows=None):
        """Full-frame buffers; only rows [row0,row1), or the rows listed in `rows`, are written."""
        if row1 is None:
            row1 = self.resy
        px = np.zeros((self.resy, self.resx, 3), dtype=np.float64)
        hit = np.full((self.resy, self.resx), -2, dtype=np.int32) if want_hits else None
        ht = np.zeros((self.resy, self.resx), dtype=np.float64) if want_hits else None
        mask = np.zeros((self.resy, self.resx), dtype=np.uint8) if want_mask else None
        ctr = (C.c_uint64 * 7)()
        outs = (_dp(px), hit.ctypes.data_as(C.POINTER(C.c_int)) if want_hits else None, _dp(ht) if want_hits else None,
                mask.ctypes.data_as(C.POINTER(C.c_ubyte)) if want_mask else None, ctr)
        if rows is not None:
            rl = np.ascontiguousarray(rows, dtype=np.int32)
            rc = load().orc_render_rows(self._h, num_samples, aperture, threads, rl.ctypes.data_as(C.POINTER(C.c_int)),
                                        len(rl), seed, flags, *outs)
        else:
            rc = load().orc_render(self._h, num_samples, aperture, threads, row0, row1, seed, flags, *outs)
        if rc != 0:
            raise RuntimeError("oracle render failed")
        return {"rgb_f64": px, "rgb_u8": px.astype(np.int32).astype(np.uint8) if want_u8 else None, "hit_id": hit, "hit_t": ht,
                "lightcam_mask": mask, "counters": dict(zip(COUNTER_NAMES, [int(x) for x in ctr]))}

    def solid(self, i):
        t, hv = C.c_int(), C.c_int()
        mat, geom, vn = np.zeros(8), np.zeros(9), np.zeros(9)
        load().orc_get_solid(self._h, i, C.byref(t), _dp(mat), _dp(geom), C.byref(hv), _dp(vn))
        return {"type": t.value, "material": mat, "geom": geom, "has_vn": hv.value, "vn": vn}

    def camera(self):
        out = np.zeros(17)
        load().orc_get_camera(self._h, _dp(out))
        return {"u": out[0:3], "v": out[3:6], "w": out[6:9], "e": out[9:12], "d": out[12], "pix_width": out[13],
                "l": out[14], "t": out[15]}

    def lights(self):
        out = np.zeros((self.n_lights, 7))
        for i in range(self.n_lights):
            load().orc_get_light(self._h, i, _dp(out[i]))
        return out

    def background(self):
        out = np.zeros(3)
        load().orc_get_background(self._h, _dp(out))
        return out

    def octree(self):
        """Pre-order dump: node (centre, radius) rows, resident-solid counts, concatenated solid ids."""
        n_ids = load().orc_dump_octree(self._h, None, None, None)
        node4 = np.zeros((self.n_nodes, 4))
        cnt = np.zeros(self.n_nodes, dtype=np.int32)
        ids = np.zeros(max(n_ids, 1), dtype=np.int32)
        load().orc_dump_octree(self._h, _dp(node4), cnt.ctypes.data_as(C.POINTER(C.c_int)),
                               ids.ctypes.data_as(C.POINTER(C.c_int)))
        return node4, cnt, ids[:n_ids]

    def lightcams(self):
        out = []
        for i in range(load().orc_lightcam_count(self._h)):
            s, l, n, r = C.c_int(), C.c_int(), C.c_int(), C.c_uint64()
            load().orc_lightcam_info(self._h, i, C.byref(s), C.byref(l), C.byref(n), C.byref(r))
            out.append({"sphere": s.value, "light": l.value, "points": n.value, "light_rays": int(r.value)})
        return out


def test_triangle(v9, d, e, min_t, max_t, vn9=None, phong=False):
    out = np.zeros(7)
    v9 = np.ascontiguousarray(v9, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    vn = np.ascontiguousarray(vn9, dtype=np.float64) if vn9 is not None else None
    t = load().orc_test_triangle(_dp(v9), _dp(vn) if vn is not None else None, 1 if phong else 0, _dp(d), _dp(e),
                                 min_t, max_t, _dp(out))
    return t, out


def test_sphere(c3, r, d, e, min_t, max_t):
    out = np.zeros(7)
    c3 = np.ascontiguousarray(c3, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    t = load().orc_test_sphere(_dp(c3), r, _dp(d), _dp(e), min_t, max_t, _dp(out))
    return t, out


def test_bound(c3, r, d, e, min_t, max_t):
    c3 = np.ascontiguousarray(c3, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    return load().orc_test_bound(_dp(c3), r, _dp(d), _dp(e), min_t, max_t)


def test_batch(kind, geom, rays, min_t, max_t):
    """kind 0 triangle (geom n x 9), 2 sphere (n x 4), 3 node bound (n x 4); rays n x 6 (origin, direction).
    Returns n x 4: t, then the hit record's unit normal (zero on a miss and for kind 3)."""
    geom = np.ascontiguousarray(geom, dtype=np.float64)
    rays = np.ascontiguousarray(rays, dtype=np.float64)
    out = np.zeros((len(rays), 4))
    load().orc_test_batch(kind, len(rays), _dp(geom), _dp(rays), min_t, max_t, _dp(out))
    return out
