"""The multi-GPU path's device pieces on ONE GPU: stripe pack / unpack kernels against numpy, the asynchronous accumulate,
and the whole dr_group pipeline (one context + one host thread per rank, double-buffered gather beside rendering) with
several ranks sharing the device (peer-copy transport; RCCL needs distinct GPUs and first runs in the driver's scaling bench)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dr():
    import dogeray_amd
    assert dogeray_amd.device_count() >= 1
    return dogeray_amd


@pytest.fixture(scope="module")
def scene(dr, synth):
    sc = dr.Scene.load(os.path.join(synth["dir"], "city_small.rts"))
    sc.build_bvh()
    return sc


W, H = 328, 200        # 41 block columns: not a multiple of 2, 3 or 8; 25 block rows


def _dev_tensor(dr, ptr, nelems):
    import torch
    from dogeray_amd import multigpu
    return torch.as_tensor(multigpu._DevArray(ptr, nelems), device=torch.device("cuda", 0))


@pytest.mark.parametrize("R", [2, 3, 8])
def test_pack_and_unpack_against_numpy(dr, scene, R):
    import torch
    s = scene.settings()
    st = dr.pack_settings13(s, 1)
    ctx = dr.Context(0).upload(scene)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 5, 1000003, 3)
    full = ctx.accum_read()                                   # [W, H, 3]
    gx, run = W // 8, 8 * H * 3
    cols = full.reshape(-1)[: gx * run].reshape(gx, run)
    stride = ((gx + R - 1) // R) * run
    stage = np.zeros((R, stride), dtype=np.int32)
    for r in range(R):
        ctx.set_stripe(R, r)
        ptr, nbytes = ctx.accum_pack_stripe(r & 1)
        ctx.synchronize()
        want = cols[r::R]
        assert nbytes == want.size * 4
        got = _dev_tensor(dr, ptr, want.size).cpu().numpy().reshape(want.shape)
        assert np.array_equal(got, want), (R, r)
        stage[r, : want.size] = want.reshape(-1)
    # unpack: ranks 1..R-1 from a staging buffer into an accumulator that holds only rank 0's columns
    ctx.set_stripe(R, 0)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 5, 1000003, 3)
    mine = ctx.accum_read()
    assert not mine.reshape(-1)[: gx * run].reshape(gx, run)[1::R].any() or R == 1
    dev = torch.from_numpy(stage).cuda()
    ctx.accum_unpack_stripes(dev.data_ptr(), stride * 4, R, 1)
    ctx.synchronize()
    assert np.array_equal(ctx.accum_read(), full), R
    ctx.close()


def test_async_accumulate_equals_blocking(dr, scene):
    s = scene.settings()
    st = dr.pack_settings13(s, 1)
    ctx = dr.Context(0).upload(scene)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 9, 1000003, 7)
    want = ctx.accum_read()
    ctx.accum_reset(W, H)
    ctx.stats_reset()
    for k, n in ((0, 2), (2, 3), (5, 1), (6, 1)):              # four batches: the third call waits for the first
        ctx.render_accumulate_async(st, W, H, s.background, 9 + k * 1000003, 1000003, n)
    ctx.synchronize()
    assert np.array_equal(ctx.accum_read(), want)
    stats = ctx.stats()
    assert stats["frames"] == 7 and stats["kernel_ms"] > 0
    ctx.close()


@pytest.mark.parametrize("ranks,every", [(1, 0), (2, 2), (3, 1), (3, 4), (8, 3)])
def test_group_on_one_device_assembles_the_frame(dr, scene, ranks, every):
    """dr_group with `ranks` contexts on device 0: the frame assembled on rank 0 is the single-context frame, bit for bit."""
    s = scene.settings()
    st = dr.pack_settings13(s, 1)
    ctx = dr.Context(0).upload(scene)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 31, 1000003, 9)
    want = ctx.accum_read()
    ctx.close()
    g = dr.Group([0] * ranks).upload(scene)
    assert g.size == ranks and not g.uses_rccl
    g.accum_reset(W, H)
    g.render_accumulate(st, W, H, s.background, 31, 1000003, 5, gather_every=every)
    g.render_accumulate(st, W, H, s.background, 31 + 5 * 1000003, 1000003, 4, gather_every=every)     # accumulates on top
    assert np.array_equal(g.accum_read(), want), (ranks, every)
    g.close()


def test_group_failure_returns_an_error_and_destroys_cleanly(dr, scene, monkeypatch):
    """A rank that fails in the middle of a render call (injected: DOGERAY_GROUP_FAIL_RANK / _BATCH): the call returns an error -- no hang: the other
    ranks are released from the rendezvous, stream waits are polls with a deadline -- and the group is destroyed cleanly.  Copy transport with three ranks
    on this device (the group stays usable), and the one-rank RCCL rehearsal (the communicator is aborted by the coordinator once every rank thread is
    back: the group says it is unusable)."""
    import time
    s = scene.settings()
    st = dr.pack_settings13(s, 1)
    monkeypatch.setenv("DOGERAY_GROUP_TIMEOUT_S", "20")
    monkeypatch.setenv("DOGERAY_GROUP_FAIL_RANK", "1")
    monkeypatch.setenv("DOGERAY_GROUP_FAIL_BATCH", "1")
    g = dr.Group([0, 0, 0]).upload(scene)
    g.accum_reset(W, H)
    t0 = time.time()
    with pytest.raises(dr.DogerayError, match="injected failure"):
        g.render_accumulate(st, W, H, s.background, 31, 1000003, 6, gather_every=2)
    assert time.time() - t0 < 15
    g.close()
    # without the injection the same group shape works (the environment is read when a group is created)
    monkeypatch.delenv("DOGERAY_GROUP_FAIL_RANK"); monkeypatch.delenv("DOGERAY_GROUP_FAIL_BATCH")
    g = dr.Group([0, 0, 0]).upload(scene)
    g.accum_reset(W, H)
    g.render_accumulate(st, W, H, s.background, 31, 1000003, 6, gather_every=2)
    g.close()
    # one rank on RCCL
    monkeypatch.setenv("DOGERAY_GROUP_TRANSPORT", "rccl")
    monkeypatch.setenv("DOGERAY_GROUP_FAIL_RANK", "0")
    monkeypatch.setenv("DOGERAY_GROUP_FAIL_BATCH", "1")
    g = dr.Group([0]).upload(scene)
    assert g.uses_rccl and g.rccl_ranks == 1
    g.accum_reset(W, H)
    t0 = time.time()
    with pytest.raises(dr.DogerayError, match="injected failure"):
        g.render_accumulate(st, W, H, s.background, 31, 1000003, 6, gather_every=2)
    assert time.time() - t0 < 15
    with pytest.raises(dr.DogerayError, match="unusable"):
        g.render_accumulate(st, W, H, s.background, 31, 1000003, 2, gather_every=2)
    g.close()


def test_native_host_application_with_gpus(dr, synth, tmp_path):
    """csrc/dogeray_main.cpp --gpus 3 (three ranks on this GPU): the image equals the one-GPU run's."""
    exe = os.path.join(ROOT, "dogeray_amd", "bin", "dogeray")
    scene = os.path.join(synth["dir"], "city_small.rts")
    outs = []
    for gpus in (1, 3):
        out = str(tmp_path / ("img%d.ppm" % gpus))
        env = dict(os.environ, DOGERAY_GROUP_DEVICES="0,0,0")
        cmd = [exe, scene, "--textures", synth["tex"], "--frames", "12", "--group", "5", "--gather-every", "2", "--out", out, "--quiet"]
        if gpus > 1:
            cmd += ["--gpus", str(gpus)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000


def test_native_host_application_pipelined_present_loop(dr, synth, tmp_path):
    """csrc/dogeray_main.cpp --group 1: one frame per displayed image, through dr_pipeline_submit / dr_pipeline_wait (frame k + 1 queued
    before frame k's image is waited for).  The last image shown is clamp(sum of all frames / count), so it must equal the image of
    a run that renders the same frames in batches of 5."""
    exe = os.path.join(ROOT, "dogeray_amd", "bin", "dogeray")
    scene = os.path.join(synth["dir"], "city_small.rts")
    outs = []
    for group in (1, 5):
        out = str(tmp_path / ("loop%d.ppm" % group))
        r = subprocess.run([exe, scene, "--textures", synth["tex"], "--frames", "13", "--group", str(group), "--out", out, "--quiet"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000


def test_stripe_gatherer_with_a_stand_in_collective(dr, scene, monkeypatch):
    """multigpu.StripeGatherer (bench.py --gpus N on RCCL) for two ranks in ONE process: torch.distributed.gather is replaced
    by a copy between the two ranks' buffers (RCCL refuses two ranks on one device), everything else -- pack kernel on the
    library's stream, event ordering against torch's stream, unpack on torch's stream, buffer alternation -- is the real path."""
    import torch
    import torch.distributed as dist
    from dogeray_amd import multigpu
    s = scene.settings()
    st = dr.pack_settings13(s, 1)
    dev = torch.device("cuda", 0)
    ref = dr.Context(0).upload(scene)
    ref.accum_reset(W, H)
    ref.render_accumulate(st, W, H, s.background, 77, 1000003, 6)
    want = ref.accum_read()
    ref.close()
    ctxs = [dr.Context(0).upload(scene) for _ in range(2)]
    mailbox = {}

    def fake_gather(tensor, gather_list=None, dst=0, group=None):
        if gather_list is None:                       # rank 1 "sends": keep a copy made on the current stream
            mailbox["rank1"] = tensor.clone()
        else:
            gather_list[0].copy_(tensor)
            gather_list[1].copy_(mailbox.pop("rank1"))

    monkeypatch.setattr(dist, "gather", fake_gather)
    gs = []
    for r, c in enumerate(ctxs):
        c.set_stripe(2, r)
        c.accum_reset(W, H)
        gs.append(multigpu.StripeGatherer(c, W, H, 2, r, dev))
    for k in range(3):                                # three batches of two frames: both pack slots and both staging buffers are reused
        for r in (1, 0):                              # rank 1 first, so that its stripe is in the mailbox when rank 0 gathers
            ctxs[r].render_accumulate_async(st, W, H, s.background, 77 + 2 * k * 1000003, 1000003, 2)
            gs[r].gather_async()
    for g in gs:
        g.finish()
    assert np.array_equal(ctxs[0].accum_read(), want)
    for c in ctxs:
        c.close()


def test_stripe_gatherer_on_rccl_with_one_rank(dr, scene):
    """The real collective library under the real buffers and streams: a one-rank RCCL process group (all a one-GPU box allows),
    StripeGatherer in rehearsal mode -- rank 0 gathers its own packed stripe (with one rank: the whole frame) through
    torch.distributed (backend nccl = RCCL) into its staging buffers, batch after batch, beside the rendering of the next batch;
    the last gathered buffer must be the final accumulator, the one before it the accumulator two frames earlier."""
    import socket
    import torch
    import torch.distributed as dist
    from dogeray_amd import multigpu
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        s = scene.settings()
        st = dr.pack_settings13(s, 1)
        ref = dr.Context(0).upload(scene)
        ref.accum_reset(W, H)
        ref.render_accumulate(st, W, H, s.background, 55, 1000003, 6)
        want = ref.accum_read()
        ref.close()
        ctx = dr.Context(0).upload(scene)
        ctx.accum_reset(W, H)
        g = multigpu.StripeGatherer(ctx, W, H, 1, 0, torch.device("cuda", 0), rehearse=True)
        for k in range(3):
            ctx.render_accumulate_async(st, W, H, s.background, 55 + 2 * k * 1000003, 1000003, 2)
            g.gather_async()
        g.finish()
        assert np.array_equal(ctx.accum_read(), want)
        n = (W // 8) * 8 * H * 3
        last = g.stage[(g.batch - 1) & 1][0][:n].cpu().numpy()
        assert np.array_equal(last, want.reshape(-1)[:n])
        ref = dr.Context(0).upload(scene)
        ref.accum_reset(W, H)
        ref.render_accumulate(st, W, H, s.background, 55, 1000003, 4)
        prev = g.stage[g.batch & 1][0][:n].cpu().numpy()
        assert np.array_equal(prev, ref.accum_read().reshape(-1)[:n])
        ref.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_native_host_application_loads_rccl(dr, synth, tmp_path):
    """`dogeray --gpus 1` with DOGERAY_GROUP_TRANSPORT=rccl: the group resolves RCCL from librccl.so, creates a one-rank communicator
    and sends a buffer to itself before rendering (the entry points dr_group uses for N > 1, as far as one GPU can exercise them)."""
    exe = os.path.join(ROOT, "dogeray_amd", "bin", "dogeray")
    scene = os.path.join(synth["dir"], "city_small.rts")
    out = str(tmp_path / "img.ppm")
    env = dict(os.environ, DOGERAY_GROUP_TRANSPORT="rccl")
    r = subprocess.run([exe, scene, "--textures", synth["tex"], "--frames", "4", "--gpus", "1", "--out", out], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RCCL send/recv" in r.stdout and os.path.getsize(out) > 1000
