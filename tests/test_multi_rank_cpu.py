"""World-size-2 (and 3) check of the row-band partition + framebuffer gather on CPU with gloo.

Each rank fills its compact row buffer from an oracle render (a stand-in for the GPU render,
which the -m gpu tests compare with the same oracle), the ranks gather to rank 0, and rank 0's
frame must equal the whole oracle frame.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import raytracer_amd as rt

RES = (40, 37)
BAND = 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    from oracle import binding as ob
    from raytracer_amd.multi import FrameGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = ob.OracleScene(rt.scene_path("tetra-3"))
        o.set_resolution(*RES)
        full = o.render(threads=1)["rgb_u8"]
        fg = FrameGather(RES[0], RES[1], BAND, world, rank, torch.device("cpu"), dist)
        rows = rt.part_row_indices(RES[1], BAND, world, rank)
        assert fg.local_rows == len(rows)
        fg.local[: len(rows)] = torch.from_numpy(full[rows])
        frame = fg.gather()
        dist.barrier()
        if rank == 0:
            ok = bool(np.array_equal(frame.numpy(), full))
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_reassembles_the_frame(tmp_path, world):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_single_rank_gather_is_a_copy():
    from raytracer_amd.multi import FrameGather
    fg = FrameGather(5, 4, 0, 1, 0, torch.device("cpu"))
    fg.local[:] = torch.arange(60, dtype=torch.uint8).reshape(4, 5, 3)
    assert torch.equal(fg.gather(), fg.local)
