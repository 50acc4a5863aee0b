"""The N > 1 path on CPU: world_size 2 and 3 over gloo (torch.distributed.run on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import CUBE_SETTINGS, ROOT, SCENES, with_settings


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_stripes_gather_to_rank0(tmp_path, world):
    # 256x200: 32 block columns (not a multiple of 3: ranks own 11/11/10) and 25 block rows
    scene = with_settings(os.path.join(SCENES, "cube.rts"), str(tmp_path / "cube.rts"), CUBE_SETTINGS.replace("256,256", "256,200"))
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_gather_worker.py"), scene, out]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert open(out).read() == "OK"


def test_owned_columns_partition():
    from dogeray_amd import multigpu
    for gx in (1, 7, 240, 241):
        for world in (1, 2, 3, 8):
            cols = sorted(c for r in range(world) for c in multigpu.owned_columns(gx, world, r))
            assert cols == list(range(gx))
