"""rt_group, the multi-GPU form of the render boundary, on what a one-GPU box can show: a group whose members are several
contexts on the same device (the partition, the per-member host threads, the gather and the row placement are the ones a
node with several GPUs runs; only the peer copies degenerate to same-device ones), and the one-member group."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members,band,staged", [(1, 16, 0), (3, 16, 0), (4, 7, 0), (8, 1, 0), (3, 5, 1)])
def test_group_frame_equals_single_context_and_oracle(oracle_frame, members, band, staged):
    """Every member writes its rows straight into the frame on the first device; with the option `staged` the parts go
    through a peer copy to the first device and are placed there, the path of devices that cannot reach each other."""
    s = rt.Scene.load("teapot-3")
    cam = s.camera(150, 133)                    # neither a multiple of the band height nor of 4 bytes per row
    o = oracle_frame("teapot-3", res=(150, 133))
    g = rt.Group([0] * members)
    g.set_option("staged", staged)
    g.upload(s)
    fr = g.render(cam, band_rows=band, flags=rt.FLAG_KEEP_ZERO_WEIGHT | rt.FLAG_TIMING)
    assert fr["stats"]["primary_launches"] >= members and fr["stats"]["primary_ms"] > 0
    assert np.array_equal(fr["rgb_u8"], o["rgb_u8"])
    assert np.array_equal(fr["hit_id"], o["hit_id"])
    assert np.abs(fr["rgb_f64"] - o["rgb_f64"]).max() <= 1e-9
    assert fr["stats"]["closest_rays"] == o["counters"]["closest_rays"]
    assert fr["stats"]["shadow_rays"] == o["counters"]["shadow_rays"]
    g.close()
    s.close()


def test_group_gather_as_rccl_send_and_receive(oracle_frame):
    """Option `rccl`: the members' parts travel as grouped ncclSend / ncclRecv to the first member, which places the rows.  A
    communicator takes a device once, so on a one-GPU box this is a group of one (the member sends to itself): the library is
    loaded, the communicator made, and the frame comes out through send, receive and placement.  A group that lists a device
    twice is told so."""
    s = rt.Scene.load("teapot-3")
    cam = s.camera(150, 133)
    o = oracle_frame("teapot-3", res=(150, 133))
    g = rt.Group([0])
    g.set_option("rccl", 1)
    g.upload(s)
    for band in (16, 5):
        fr = g.render(cam, band_rows=band, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        assert np.array_equal(fr["rgb_u8"], o["rgb_u8"])
        assert np.array_equal(fr["hit_id"], o["hit_id"])
        assert np.abs(fr["rgb_f64"] - o["rgb_f64"]).max() <= 1e-9
    g.set_option("rccl", 0)
    fr = g.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    assert np.array_equal(fr["rgb_u8"], o["rgb_u8"])
    g.close()
    g2 = rt.Group([0, 0])
    with pytest.raises(rt.RtError, match="twice"):
        g2.set_option("rccl", 1)
    g2.close()
    s.close()


def test_group_with_light_cameras_and_device_output(oracle_frame):
    import torch
    s = rt.Scene.load("glass3")
    cam = s.camera(160, 160)
    o = oracle_frame("glass3", res=(160, 160))
    g = rt.Group([0, 0])
    g.upload(s)
    fr = g.render(cam, num_samples=1)
    assert np.array_equal(fr["rgb_u8"], o["rgb_u8"])
    assert np.array_equal(fr["lightcam_mask"], o["lightcam_mask"])
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    assert np.array_equal(u8.cpu().numpy(), o["rgb_u8"])
    assert st["closest_rays"] == o["counters"]["closest_rays"]
    with pytest.raises(rt.RtError):
        rt.Group([0, 99])
    g.close()
    s.close()


def test_cli_driver_with_gpus_flag(tmp_path):
    """`rt_driver --gpus 1` is the one-GPU path; asking for more GPUs than the box has is a clear error."""
    import os
    import subprocess
    driver = os.path.join(os.path.dirname(rt.LIB_PATH), "bin", "rt_driver")
    out = tmp_path / "t.ppm"
    r = subprocess.run([driver, "--gpus", "1", rt.scene_path("tetra-3"), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ref = tmp_path / "ref.ppm"
    r2 = subprocess.run([driver, rt.scene_path("tetra-3"), str(ref)], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0
    assert out.read_bytes() == ref.read_bytes()
    r4 = subprocess.run([driver, "--gpus", "1", "--rccl", rt.scene_path("tetra-3"), str(out)], capture_output=True, text=True, timeout=120)
    assert r4.returncode == 0, r4.stderr
    assert out.read_bytes() == ref.read_bytes()
    r3 = subprocess.run([driver, "--gpus", "64", rt.scene_path("tetra-3"), str(out)], capture_output=True, text=True, timeout=120)
    assert r3.returncode != 0 and "GPU" in r3.stderr
