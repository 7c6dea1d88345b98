"""rt_render_noalloc, the entry point for callers that cannot allocate device memory: its frames against
rt_render_async's on the same inputs, through the C ABI.

Two calls in a row share the library's static scratch array, the second with a smaller frame, so stale contents of
the first call would show.  Integer outputs (8-bit image, hit ids, ray counts) must be identical, the double frame
within 1e-9 of 255 (the order of a pixel's atomic FP64 additions is free in both entry points).
"""
import numpy as np
import pytest
import torch

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def _device_frame(cam):
    return (torch.zeros((cam.resy, cam.resx, 3), dtype=torch.float64, device="cuda"),
            torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda"),
            torch.full((cam.resy, cam.resx), -2, dtype=torch.int32, device="cuda"),
            torch.zeros((cam.resy, cam.resx), dtype=torch.uint8, device="cuda"))


def _both(gpu, cam, **kw):
    a, b = _device_frame(cam), _device_frame(cam)
    st_a = gpu.render_device(cam, rgb_f64_ptr=a[0].data_ptr(), rgb_u8_ptr=a[1].data_ptr(), hit_id_ptr=a[2].data_ptr(),
                             mask_ptr=a[3].data_ptr(), **kw)
    st_b = gpu.render_noalloc(cam, rgb_f64_ptr=b[0].data_ptr(), rgb_u8_ptr=b[1].data_ptr(), hit_id_ptr=b[2].data_ptr(),
                              mask_ptr=b[3].data_ptr(), **kw)
    torch.cuda.synchronize()
    assert torch.equal(a[1], b[1]), "8-bit frames differ"
    assert torch.equal(a[2], b[2]), "hit ids differ"
    assert torch.equal(a[3], b[3]), "light-camera masks differ"
    assert float((a[0] - b[0]).abs().max()) <= 1e-9
    for k in ("closest_rays", "shadow_rays", "rays_per_generation", "hits_per_generation"):
        assert st_a[k] == st_b[k], k
    assert st_b["queue_overflow"] == 0
    return st_b


def test_noalloc_equals_render_async_on_two_calls_in_a_row():
    gpu = rt.Renderer(0)
    nbytes = rt.noalloc_scratch_bytes()
    assert nbytes == 128 << 20
    s = rt.Scene.load("refract")            # transparent material: both kinds of secondary ray, 64- and 128-byte hits
    gpu.upload(s)
    big = _both(gpu, s.camera(320, 300), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    small = _both(gpu, s.camera(96, 70), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    assert small["closest_rays"] < big["closest_rays"]
    # supersampled, in row bands, another scene: same context, same scratch
    s2 = rt.Scene.load("balls")
    gpu.upload(s2)
    _both(gpu, s2.camera(200, 200), num_samples=2, seed=3)
    _both(gpu, s2.camera(200, 200), band_rows=16, n_parts=3, part=1)
    # it allocates nothing: the device's free memory is the same before and after calls of different sizes (the
    # output buffers are made first; hipMemGetInfo sees every hipMalloc of the process, the library's included)
    cams = [s2.camera(200, 200), s2.camera(64, 48), s2.camera(300, 200)]
    outs = [_device_frame(c) for c in cams]
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for c, o in zip(cams, outs):
        gpu.render_noalloc(c, rgb_f64_ptr=o[0].data_ptr(), rgb_u8_ptr=o[1].data_ptr(), hit_id_ptr=o[2].data_ptr())
    assert torch.cuda.mem_get_info()[0] == free_before
    # light cameras: refused until they exist (building them allocates), then served from the scratch as well
    s3 = rt.Scene.load("glass3")
    gpu.upload(s3)
    cam3 = s3.camera(150, 150)
    out = _device_frame(cam3)
    with pytest.raises(rt.RtError) as e:
        gpu.render_noalloc(cam3, rgb_u8_ptr=out[1].data_ptr())
    assert e.value.code == -5
    _both(gpu, cam3)
    # limits are errors: a frame whose accumulation frame alone exceeds the scratch, host-side flags that allocate
    with pytest.raises(rt.RtError) as e:
        gpu.render_noalloc(s3.camera(4096, 4096), rgb_u8_ptr=out[1].data_ptr(), flags=rt.FLAG_NO_LIGHTCAM)
    assert e.value.code == -4 and "static scratch" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        gpu.render_noalloc(cam3, rgb_u8_ptr=out[1].data_ptr(), flags=rt.FLAG_TIMING)
    assert e.value.code == -5
    # a second context on the same GPU uses the same array, one call after the other
    other = rt.Renderer(0)
    other.upload(s2)
    _both(other, s2.camera(64, 64))
    other.close()
    gpu.close()
    for x in (s, s2, s3):
        x.close()
