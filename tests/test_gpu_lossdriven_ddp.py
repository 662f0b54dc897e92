"""The loss-driven RetinaNet step behind the data-parallel wrapper: two ranks on one GPU (gloo), gradients averaged by the parameter
hooks == the mean of the two ranks' dense single-process gradients (cosine; the backbone is not run-to-run reproducible)."""
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_take_the_loss_driven_step_and_average_their_gradients():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_ddp_gpu_retina_script.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-2000:], e[-3000:])
    assert "cos " in outs[0][0]
