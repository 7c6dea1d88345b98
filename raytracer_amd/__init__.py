"""Python host for the MI355X render path.

Thin ctypes binding over the C ABI of include/raytracer_amd.h (libraytracer_amd.so).
There is no CPU path here: every render call goes to the HIP library, and a missing
library or a missing GPU is an error, never a fallback.

    scene = Scene.load("teapot-3.nff")           # host front end (parse, triangulate, octree)
    r = Renderer(device=0); r.upload(scene)
    frame = r.render(scene.camera(), num_samples=1)   # dict: rgb_f64, rgb_u8, hit_id, stats
"""
import ctypes as C
import gzip
import os
import tempfile

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("RAYTRACER_AMD_LIB") or os.path.join(_PKG, "libraytracer_amd.so")

RT_OK = 0
FLAG_PHONG_NORMALS = 1
FLAG_NO_CULL = 2
FLAG_KEEP_ZERO_WEIGHT = 4
FLAG_TIMING = 8
FLAG_NO_LIGHTCAM = 16
SPH_TRANSPARENT = 0x80000000
PRIM_SPHERE = 1
PRIM_HAS_VNORMALS = 2


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("raytracer_amd error %d: %s" % (code, msg))
        self.code = code


class Material(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("r", "g", "b", "Kd", "Ks", "Shine", "T", "ior")]


class Light(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("color", C.c_double * 3), ("intensity", C.c_double), ("_pad", C.c_double)]


class Node(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("skip", C.c_int32),
                ("tri_begin", C.c_int32), ("tri_count", C.c_int32), ("sph_begin", C.c_int32),
                ("sph_count", C.c_int32), ("depth", C.c_int32), ("_pad", C.c_int32 * 2)]


class TriRec(C.Structure):
    _fields_ = [("v0", C.c_double * 3), ("ab", C.c_double * 3), ("ac", C.c_double * 3),
                ("prim", C.c_int32), ("_pad0", C.c_int32), ("_pad1", C.c_double * 2)]


class SphRec(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("n_nodes", C.c_int32),
                ("tri", C.POINTER(TriRec)), ("n_tri_slots", C.c_int32),
                ("sph", C.POINTER(SphRec)), ("n_sph_slots", C.c_int32),
                ("sph_prim", C.POINTER(C.c_uint32)),
                ("n_prims", C.c_int32),
                ("prim_flags", C.POINTER(C.c_uint32)),
                ("prim_material", C.POINTER(C.c_int32)),
                ("prim_normals", C.POINTER(C.c_double)),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_int32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_int32),
                ("background", C.c_double * 3),
                ("lightcam_samples", C.c_int32), ("_pad", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3), ("e", C.c_double * 3),
                ("d", C.c_double), ("l", C.c_double), ("t", C.c_double), ("pix_width", C.c_double),
                ("resx", C.c_int32), ("resy", C.c_int32)]


class RenderOpts(C.Structure):
    _fields_ = [("num_samples", C.c_int32), ("aperture", C.c_double), ("flags", C.c_uint32), ("seed", C.c_uint32),
                ("band_rows", C.c_int32), ("n_parts", C.c_int32), ("part", C.c_int32), ("_pad", C.c_int32)]


class FrameOut(C.Structure):
    _fields_ = [("rgb_f64", C.c_void_p), ("rgb_u8", C.c_void_p), ("hit_id", C.c_void_p),
                ("lightcam_mask", C.c_void_p), ("on_device", C.c_int32), ("_pad", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("lightcam_rays", C.c_uint64),
                ("rays_per_generation", C.c_uint64 * 6), ("hits_per_generation", C.c_uint64 * 6),
                ("gpu_ms", C.c_double), ("closest_ms", C.c_double), ("shadow_ms", C.c_double),
                ("other_ms", C.c_double), ("closest_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
                ("batches", C.c_int32), ("queue_overflow", C.c_int32),
                ("primary_ms", C.c_double), ("primary_launches", C.c_uint64),
                ("pair_ms", C.c_double), ("pair_launches", C.c_uint64)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


# Every symbol the header declares, with its signature; load() checks that all are exported.
_SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_build_id": (C.c_char_p, []),
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rt_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rt_destroy": (C.c_int, [C.c_void_p]),
    "rt_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rt_part_rows": (C.c_int, [C.POINTER(Camera), C.POINTER(RenderOpts), C.POINTER(C.c_int32)]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.POINTER(FrameOut), C.POINTER(Stats)]),
    "rt_render_async": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.POINTER(FrameOut)]),
    "rt_sync": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rt_set_batch": (C.c_int, [C.c_void_p, C.c_int64]),
    "rt_set_memory_budget": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_render_noalloc": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.POINTER(FrameOut),
                                    C.POINTER(Stats)]),
    "rt_noalloc_scratch_bytes": (C.c_int, [C.POINTER(C.c_size_t)]),
    "rt_debug_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "rt_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_group_destroy": (C.c_int, [C.c_void_p]),
    "rt_group_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "rt_group_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rt_group_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.POINTER(FrameOut), C.POINTER(Stats)]),
    "rt_group_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "rt_debug_intersect": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "rth_scene_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rth_scene_free": (C.c_int, [C.c_void_p]),
    "rth_scene_desc": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rth_scene_camera": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Camera)]),
    "rth_scene_resolution": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rth_scene_prim_geometry": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "rth_write_ppm_f64": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.c_int32, C.c_int32]),
    "rth_write_ppm_u8": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    "rth_parse_args": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int32), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
}

_lib = None


def load():
    """Loads libraytracer_amd.so and binds every declared entry point (no GPU needed for that)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(-2, "%s is missing: build it with `python -m raytracer_amd.build`; there is no fallback path"
                      % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != RT_OK:
        raise RtError(rc, load().rt_last_error().decode("utf-8", "replace"))


def scene_path(name):
    """Path of a bundled scene (tests/golden/scenes/<name>.nff.gz), unpacked to a temp dir."""
    if os.path.exists(name):
        return name
    base = name[:-4] if name.endswith(".nff") else name
    src = os.path.join(_ROOT, "tests", "golden", "scenes", base + ".nff.gz")
    if not os.path.exists(src):
        raise FileNotFoundError("no such scene: %s" % name)
    out_dir = os.path.join(tempfile.gettempdir(), "raytracer_amd_scenes_%d" % os.getuid())
    os.makedirs(out_dir, exist_ok=True)
    dst = os.path.join(out_dir, base + ".nff")
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        with gzip.open(src, "rb") as f:
            data = f.read()
        tmp = dst + ".%d.tmp" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, dst)
    return dst


def _np_view(ptr, n, dtype):
    if n == 0:
        return np.zeros((0,), dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype)


class Scene:
    """Host-side scene: parsed .nff, triangulated primitives and the flattened octree."""

    def __init__(self, handle):
        self._h = handle
        self.desc = SceneDesc()
        _check(load().rth_scene_desc(self._h, C.byref(self.desc)))

    @classmethod
    def load(cls, name_or_path):
        h = C.c_void_p()
        _check(load().rth_scene_load(scene_path(name_or_path).encode(), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            load().rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def resolution(self):
        rx, ry = C.c_int32(), C.c_int32()
        _check(load().rth_scene_resolution(self._h, C.byref(rx), C.byref(ry)))
        return rx.value, ry.value

    def camera(self, resx=0, resy=0):
        cam = Camera()
        _check(load().rth_scene_camera(self._h, resx, resy, C.byref(cam)))
        return cam

    def prim_geometry(self, prim):
        out = (C.c_double * 9)()
        _check(load().rth_scene_prim_geometry(self._h, prim, out))
        return np.array(out)

    # numpy views of the flattened arrays (valid while the scene lives)
    def nodes(self):
        d = self.desc
        dt = np.dtype([("center", "<f8", 3), ("radius", "<f8"), ("skip", "<i4"), ("tri_begin", "<i4"),
                       ("tri_count", "<i4"), ("sph_begin", "<i4"), ("sph_count", "<i4"), ("depth", "<i4"),
                       ("_pad", "<i4", 2)])
        raw = np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_uint8)), shape=(d.n_nodes * 64,))
        return raw.view(dt)

    def tri_records(self):
        d = self.desc
        dt = np.dtype([("v0", "<f8", 3), ("ab", "<f8", 3), ("ac", "<f8", 3), ("prim", "<i4"), ("_p0", "<i4"),
                       ("_p1", "<f8", 2)])
        if d.n_tri_slots == 0:
            return np.zeros((0,), dtype=dt)
        raw = np.ctypeslib.as_array(C.cast(d.tri, C.POINTER(C.c_uint8)), shape=(d.n_tri_slots * 96,))
        return raw.view(dt)

    def sph_records(self):
        d = self.desc
        if d.n_sph_slots == 0:
            return np.zeros((0, 4)), np.zeros((0,), dtype=np.uint32)
        rec = np.ctypeslib.as_array(C.cast(d.sph, C.POINTER(C.c_double)), shape=(d.n_sph_slots, 4))
        prim = np.ctypeslib.as_array(d.sph_prim, shape=(d.n_sph_slots,))
        return rec, prim

    def prim_arrays(self):
        d = self.desc
        n = d.n_prims
        flags = np.ctypeslib.as_array(d.prim_flags, shape=(n,))
        mat = np.ctypeslib.as_array(d.prim_material, shape=(n,))
        nrm = np.ctypeslib.as_array(d.prim_normals, shape=(n, 12))
        return flags, mat, nrm

    def materials(self):
        d = self.desc
        return np.ctypeslib.as_array(C.cast(d.materials, C.POINTER(C.c_double)), shape=(d.n_materials, 8))

    def lights(self):
        d = self.desc
        if d.n_lights == 0:
            return np.zeros((0, 8))
        return np.ctypeslib.as_array(C.cast(d.lights, C.POINTER(C.c_double)), shape=(d.n_lights, 8))


def build_id():
    return load().rt_build_id().decode()


def noalloc_scratch_bytes():
    n = C.c_size_t()
    _check(load().rt_noalloc_scratch_bytes(C.byref(n)))
    return n.value


def part_rows(cam, band_rows=0, n_parts=1, part=0):
    opts = RenderOpts(num_samples=1, band_rows=band_rows, n_parts=n_parts, part=part)
    rows = C.c_int32()
    _check(load().rt_part_rows(C.byref(cam), C.byref(opts), C.byref(rows)))
    return rows.value


def part_row_indices(resy, band_rows=0, n_parts=1, part=0):
    """Global row numbers, ascending, of the rows that `part` renders."""
    band = band_rows if band_rows > 0 else resy
    rows = []
    n_bands = (resy + band - 1) // band
    for b in range(part, n_bands, max(n_parts, 1)):
        rows.extend(range(b * band, min((b + 1) * band, resy)))
    return np.array(rows, dtype=np.int64)


class Renderer:
    """One GPU context (rt_ctx)."""

    created = 0      # contexts made by this process so far (a process that has made one has initialised the GPU)

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        _check(load().rt_create(device, C.byref(self._ctx)))
        self.device = device
        Renderer.created += 1

    def close(self):
        if self._ctx:
            load().rt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene):
        _check(load().rt_upload_scene(self._ctx, C.byref(scene.desc)))

    def set_batch(self, primaries):
        _check(load().rt_set_batch(self._ctx, primaries))

    def set_memory_budget(self, nbytes):
        _check(load().rt_set_memory_budget(self._ctx, nbytes))

    def set_option(self, name, value):
        """Experiment switch of this context (rt_debug_set_option)."""
        _check(load().rt_debug_set_option(self._ctx, name.encode(), int(value)))

    def render_noalloc(self, cam, rgb_f64_ptr=0, rgb_u8_ptr=0, hit_id_ptr=0, mask_ptr=0, num_samples=1, aperture=0.0,
                       flags=0, seed=0, band_rows=0, n_parts=1, part=0):
        """rt_render_noalloc: device output buffers, work memory from the library's static scratch."""
        opts = self._opts(num_samples, aperture, flags, seed, band_rows, n_parts, part)
        out = FrameOut(rgb_f64=rgb_f64_ptr or None, rgb_u8=rgb_u8_ptr or None, hit_id=hit_id_ptr or None,
                       lightcam_mask=mask_ptr or None, on_device=1)
        stats = Stats()
        _check(load().rt_render_noalloc(self._ctx, C.byref(cam), C.byref(opts), C.byref(out), C.byref(stats)))
        return stats.as_dict()

    @staticmethod
    def _opts(num_samples, aperture, flags, seed, band_rows, n_parts, part):
        return RenderOpts(num_samples=num_samples, aperture=aperture, flags=flags, seed=seed,
                          band_rows=band_rows, n_parts=n_parts, part=part)

    def render(self, cam, num_samples=1, aperture=0.0, flags=0, seed=0, band_rows=0, n_parts=1, part=0,
               want=("rgb_f64", "rgb_u8", "hit_id", "lightcam_mask")):
        """Renders into host numpy arrays.  Returns a dict with the requested arrays and `stats`."""
        opts = self._opts(num_samples, aperture, flags, seed, band_rows, n_parts, part)
        rows = C.c_int32()
        _check(load().rt_part_rows(C.byref(cam), C.byref(opts), C.byref(rows)))
        n = rows.value * cam.resx
        bufs = {}
        out = FrameOut(on_device=0)
        if "rgb_f64" in want:
            bufs["rgb_f64"] = np.zeros((rows.value, cam.resx, 3), dtype=np.float64)
            out.rgb_f64 = bufs["rgb_f64"].ctypes.data
        if "rgb_u8" in want:
            bufs["rgb_u8"] = np.zeros((rows.value, cam.resx, 3), dtype=np.uint8)
            out.rgb_u8 = bufs["rgb_u8"].ctypes.data
        if "hit_id" in want:
            bufs["hit_id"] = np.full((rows.value, cam.resx), -2, dtype=np.int32)
            out.hit_id = bufs["hit_id"].ctypes.data
        if "lightcam_mask" in want:
            bufs["lightcam_mask"] = np.zeros((rows.value, cam.resx), dtype=np.uint8)
            out.lightcam_mask = bufs["lightcam_mask"].ctypes.data
        stats = Stats()
        _check(load().rt_render(self._ctx, C.byref(cam), C.byref(opts), C.byref(out), C.byref(stats)))
        assert n == rows.value * cam.resx
        bufs["stats"] = stats.as_dict()
        return bufs

    def render_device(self, cam, rgb_f64_ptr=0, rgb_u8_ptr=0, hit_id_ptr=0, mask_ptr=0, num_samples=1, aperture=0.0,
                      flags=0, seed=0, band_rows=0, n_parts=1, part=0, sync=True):
        """Renders into device buffers given as raw pointers (e.g. torch tensor data_ptr())."""
        opts = self._opts(num_samples, aperture, flags, seed, band_rows, n_parts, part)
        out = FrameOut(rgb_f64=rgb_f64_ptr or None, rgb_u8=rgb_u8_ptr or None, hit_id=hit_id_ptr or None,
                       lightcam_mask=mask_ptr or None, on_device=1)
        _check(load().rt_render_async(self._ctx, C.byref(cam), C.byref(opts), C.byref(out)))
        if sync:
            return self.sync()
        return None

    def debug_intersect(self, kind, geom, rays, min_t, max_t):
        """Runs the device's primitive tests on (primitive, ray) pairs; see rt_debug_intersect.  Returns n x 4."""
        geom = np.ascontiguousarray(geom, dtype=np.float64)
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        out = np.zeros((len(rays), 4))
        dp = C.POINTER(C.c_double)
        _check(load().rt_debug_intersect(self._ctx, kind, len(rays), geom.ctypes.data_as(dp), rays.ctypes.data_as(dp),
                                         min_t, max_t, out.ctypes.data_as(dp)))
        return out

    def sync(self):
        stats = Stats()
        _check(load().rt_sync(self._ctx, C.byref(stats)))
        return stats.as_dict()


class Group:
    """Several GPUs of one node behind one render call (rt_group): one context per listed device."""

    def __init__(self, devices):
        self._g = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        _check(load().rt_group_create(ids, len(devices), C.byref(self._g)))
        self.devices = list(devices)

    def close(self):
        if self._g:
            load().rt_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene):
        _check(load().rt_group_upload_scene(self._g, C.byref(scene.desc)))

    def set_option(self, name, value):
        _check(load().rt_group_set_option(self._g, name.encode(), int(value)))

    def render(self, cam, num_samples=1, aperture=0.0, flags=0, seed=0, band_rows=16,
               want=("rgb_f64", "rgb_u8", "hit_id", "lightcam_mask")):
        """Renders the whole frame into host numpy arrays (frame order)."""
        opts = RenderOpts(num_samples=num_samples, aperture=aperture, flags=flags, seed=seed, band_rows=band_rows, n_parts=1, part=0)
        bufs = {}
        out = FrameOut(on_device=0)
        shapes = {"rgb_f64": ((cam.resy, cam.resx, 3), np.float64), "rgb_u8": ((cam.resy, cam.resx, 3), np.uint8),
                  "hit_id": ((cam.resy, cam.resx), np.int32), "lightcam_mask": ((cam.resy, cam.resx), np.uint8)}
        for name in want:
            shape, dt = shapes[name]
            bufs[name] = np.zeros(shape, dtype=dt)
            setattr(out, name, bufs[name].ctypes.data)
        stats = Stats()
        _check(load().rt_group_render(self._g, C.byref(cam), C.byref(opts), C.byref(out), C.byref(stats)))
        bufs["stats"] = stats.as_dict()
        return bufs

    def render_device(self, cam, rgb_u8_ptr=0, rgb_f64_ptr=0, num_samples=1, aperture=0.0, flags=0, seed=0, band_rows=16):
        """Renders into frame-order device buffers on the group's first device."""
        opts = RenderOpts(num_samples=num_samples, aperture=aperture, flags=flags, seed=seed, band_rows=band_rows, n_parts=1, part=0)
        out = FrameOut(rgb_f64=rgb_f64_ptr or None, rgb_u8=rgb_u8_ptr or None, on_device=1)
        stats = Stats()
        _check(load().rt_group_render(self._g, C.byref(cam), C.byref(opts), C.byref(out), C.byref(stats)))
        return stats.as_dict()


def write_ppm(path, rgb):
    """Writes a binary PPM the way the reference does ((int) truncation of value*255 doubles)."""
    a = np.ascontiguousarray(rgb)
    h, w = a.shape[0], a.shape[1]
    if a.dtype == np.uint8:
        _check(load().rth_write_ppm_u8(path.encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))
    else:
        a = a.astype(np.float64, copy=False)
        _check(load().rth_write_ppm_f64(path.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), w, h))
