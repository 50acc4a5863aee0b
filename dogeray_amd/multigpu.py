"""Framebuffer tiling across GPUs and the gather of tiles to rank 0.

The reference is single-GPU.  Pixels are independent (each thread writes only its own int3,
kernel.cu K:1010,1083-1085) and the RNG seed depends only on (x, y, frame) (K:1065), so a frame
rendered in pieces is bit-identical to the frame rendered whole.  Rank r of R renders the 8-pixel
block columns bx with bx % R == r (interleaved for load balance).  Because the framebuffer is
column-major (index x*H + y, K:1006), one block column is one contiguous run of 8*H*3 int32, so a
rank's share is a strided view [gx/R, 8*H*3] that is packed, gathered with one RCCL gather (7
point-to-point transfers into rank 0 over xGMI on an 8-GPU node) and scattered back on rank 0.

torch / torch.distributed are used only as plumbing (device tensors over the library's own
accumulator memory, process-group collectives); backend "nccl" is RCCL on ROCm, "gloo" on CPU.
"""
import numpy as np


def owned_columns(gx, world, rank):
    """Block columns (8 pixels wide) rendered by `rank`."""
    return list(range(rank, gx, world))


class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, nelems):
        self.__cuda_array_interface__ = {"shape": (nelems,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def accumulator_tensor(ctx, device):
    """torch int32 tensor aliasing the context's on-device accumulator (no copy)."""
    import torch
    ptr, nbytes = ctx.accum_device_ptr()
    return torch.as_tensor(_DevArray(ptr, nbytes // 4), device=device)


def gather_frame(acc, W, H, world, rank, group=None):
    """acc: this rank's int32[W*H*3] accumulator (zeros outside its block columns).
    Returns the assembled int32[W*H*3] frame on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    gx = W // 8
    run = 8 * H * 3
    cols = acc[: gx * run].view(gx, run)
    if world == 1:
        return acc
    per = (gx + world - 1) // world                     # ranks own per or per-1 columns
    mine = cols[rank::world]
    pack = torch.zeros((per, run), dtype=acc.dtype, device=acc.device)
    pack[: mine.shape[0]] = mine
    if rank == 0:
        parts = [torch.empty_like(pack) for _ in range(world)]
        dist.gather(pack, gather_list=parts, dst=0, group=group)
        full = torch.zeros_like(acc)
        fcols = full[: gx * run].view(gx, run)
        for r in range(world):
            n = len(range(r, gx, world))
            fcols[r::world] = parts[r][:n]
        return full
    dist.gather(pack, gather_list=None, dst=0, group=group)
    return None


def gather_frame_numpy(acc_np, W, H, world, rank, group=None):
    """Same as gather_frame for host arrays (used with the gloo backend in CPU tests)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(acc_np).reshape(-1))
    out = gather_frame(t, W, H, world, rank, group)
    return None if out is None else out.numpy().reshape(W, H, 3)
