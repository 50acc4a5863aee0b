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


class FrameGatherer:
    """gather_frame with its buffers allocated once (the bench gathers after every launch: at 8 ranks a
    launch lasts a few ms, so 25 MB of fresh zero-filled buffers per gather would show)."""

    def __init__(self, acc, W, H, world, rank, group=None):
        import torch
        self.W, self.H, self.world, self.rank, self.group = W, H, world, rank, group
        self.gx = W // 8
        self.run = 8 * H * 3
        self.per = (self.gx + world - 1) // world           # ranks own per or per-1 columns
        self.pack = torch.zeros((self.per, self.run), dtype=acc.dtype, device=acc.device)
        self.parts = [torch.empty_like(self.pack) for _ in range(world)] if rank == 0 and world > 1 else None
        self.full = torch.zeros_like(acc) if rank == 0 and world > 1 else None

    def gather(self, acc):
        import torch.distributed as dist
        if self.world == 1:
            return acc
        cols = acc[: self.gx * self.run].view(self.gx, self.run)
        mine = cols[self.rank::self.world]
        self.pack[: mine.shape[0]] = mine
        if self.rank != 0:
            dist.gather(self.pack, gather_list=None, dst=0, group=self.group)
            return None
        dist.gather(self.pack, gather_list=self.parts, dst=0, group=self.group)
        fcols = self.full[: self.gx * self.run].view(self.gx, self.run)
        for r in range(self.world):
            n = len(range(r, self.gx, self.world))
            fcols[r::self.world] = self.parts[r][:n]
        return self.full


class StripeGatherer:
    """The device path of bench.py --gpus N: the library packs this rank's stripe (dr_accum_pack_stripe, on the context's
    own stream), torch.distributed gathers the packed buffers to rank 0 (backend nccl = RCCL: R-1 point-to-point
    transfers into rank 0), and the library writes them into rank 0's accumulator (dr_accum_unpack_stripes, queued on
    torch's stream behind the gather).  Nothing waits on the host: events order the three stages, two pack buffers and
    two staging buffers alternate, so the gather of one batch runs beside the rendering of the next."""

    def __init__(self, ctx, W, H, world, rank, device, group=None, rehearse=False):
        """rehearse: run pack -> gather even for world == 1 (rank 0 gathers its own stripe into its staging buffer; nothing is
        written back: with one rank the stripe is the whole frame, which the next batch is already being rendered into):
        the only way to put the real collective library under these buffers and streams on a one-GPU box."""
        import torch
        self.ctx, self.W, self.H, self.world, self.rank, self.group, self.device = ctx, W, H, world, rank, group, device
        self.rehearse = rehearse
        gx = W // 8
        run = 8 * H * 3
        self.stride = ((gx + world - 1) // world) * run              # int32 per rank: the largest stripe (rank 0's); a multiple of 4
        self.lib_stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=device)
        self.coll_stream = torch.cuda.Stream(device=device)          # the collective and the unpack run here (a real stream: its handle is not 0)
        self.stage = [torch.empty((world, self.stride), dtype=torch.int32, device=device) for _ in range(2)] if rank == 0 else None
        self.done = [None, None]                                     # event: the gather that read pack buffer `slot` has finished
        self.batch = 0

    def gather_async(self):
        """Queue pack -> gather -> unpack for the frames rendered so far; returns at once."""
        import torch
        import torch.distributed as dist
        if self.world == 1 and not self.rehearse:
            return
        slot = self.batch & 1
        self.batch += 1
        cur = self.coll_stream
        if self.done[slot] is not None:
            self.lib_stream.wait_event(self.done[slot])              # the buffer is packed again only after its last gather
        ptr, _ = self.ctx.accum_pack_stripe(slot)
        packed = torch.as_tensor(_DevArray(ptr, self.stride), device=self.device)
        ev = torch.cuda.Event()
        ev.record(self.lib_stream)
        cur.wait_event(ev)                                           # the gather starts when the stripe is packed
        with torch.cuda.stream(cur):                                 # torch.distributed orders a collective against the current stream
            if self.rank != 0:
                dist.gather(packed, gather_list=None, dst=0, group=self.group)
            else:
                stage = self.stage[slot]
                dist.gather(packed, gather_list=[stage[r] for r in range(self.world)], dst=0, group=self.group)
                self.ctx.accum_unpack_stripes(stage.data_ptr(), self.stride * 4, self.world, 1, stream_ptr=cur.cuda_stream)
        self.done[slot] = torch.cuda.Event()
        self.done[slot].record(cur)

    def finish(self):
        import torch
        self.ctx.synchronize()
        torch.cuda.synchronize(self.device)


def gather_frame(acc, W, H, world, rank, group=None):
    """acc: this rank's int32[W*H*3] accumulator (zeros outside its block columns).
    Returns the assembled int32[W*H*3] frame on rank 0, None elsewhere."""
    return FrameGatherer(acc, W, H, world, rank, group).gather(acc)


def gather_frame_numpy(acc_np, W, H, world, rank, group=None):
    """Same as gather_frame for host arrays (used with the gloo backend in CPU tests)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(acc_np).reshape(-1))
    out = gather_frame(t, W, H, world, rank, group)
    return None if out is None else out.numpy().reshape(W, H, 3)
