"""Worker for test_multigpu_cpu.py: one rank of a gloo process group.

Each rank renders ITS block columns of the frame (with the CPU oracle standing in for a GPU, since
this runs without one), the stripes are gathered to rank 0 with dogeray_amd.multigpu.gather_frame
exactly as bench.py does on RCCL, and rank 0 checks the assembled frame against the whole frame.
"""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dogeray_amd import multigpu   # noqa: E402
from oracle import orc             # noqa: E402


def main():
    scene_path, out_path = sys.argv[1], sys.argv[2]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s = orc.Scene(scene_path)
    s.build_bvh()
    st = s.settings()
    W, H = st.width, st.height
    s13 = orc.settings13(st, 1)
    ok = True
    for seed in (11, 12):
        mine, _ = s.render(s13, W, H, st.background, seed, nthreads=2, col_mod=world, col_rem=rank)
        assert multigpu.owned_columns(W // 8, world, rank) == [c for c in range(W // 8) if c % world == rank]
        full = multigpu.gather_frame_numpy(mine, W, H, world, rank)
        if rank == 0:
            want, _ = s.render(s13, W, H, st.background, seed, nthreads=2)
            ok = ok and np.array_equal(full, want)
        else:
            assert full is None
    # the bench's form: one FrameGatherer, its buffers reused for every gather (two different frames through the same buffers)
    import torch
    g = None
    for seed in (21, 22):
        mine, _ = s.render(s13, W, H, st.background, seed, nthreads=2, col_mod=world, col_rem=rank)
        acc = torch.from_numpy(np.ascontiguousarray(mine).reshape(-1))
        if g is None:
            g = multigpu.FrameGatherer(acc, W, H, world, rank)
        full = g.gather(acc)
        if rank == 0:
            want, _ = s.render(s13, W, H, st.background, seed, nthreads=2)
            ok = ok and np.array_equal(full.numpy().reshape(W, H, 3), want)
    if rank == 0:
        with open(out_path, "w") as f:
            f.write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
