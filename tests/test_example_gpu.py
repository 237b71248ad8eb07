"""GPU: scripts/train_example.py end to end -- the reference launcher's flow (scripts/mainMultiModal.py:150-235: data manager,
build_model, optimizer, scheduler, Image3MEngine, engine.run, test) on a generated RGBNT201-layout JPEG tree, on one rank and
on two ranks started by ieee_amd.dist.launch from a plain `python` command (the two ranks share this box's GPU over gloo; on a
node the same command runs one rank per GPU over RCCL).  What only this test covers: the rank-sharded loader with its prefetch
thread (epoch order broadcast from rank 0 at every epoch), the data-parallel train loop over several epochs and the sharded
evaluation, all in one job."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus, extra_env=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GPU_MAX_HW_QUEUES"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_example.py"), "--synthetic", "8", "--epochs", "2",
                        "--batch", "16", "--workers", "2", "--gpus", str(gpus), "--lr", "1e-3"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    out = r.stdout.decode()
    m = re.search(r"FINAL mAP ([0-9.]+) after 2 epoch\(s\) on (\d+) rank\(s\), (\d+) train batches per epoch", out)
    assert m, out[-2000:]
    return float(m.group(1)), int(m.group(2)), int(m.group(3)), out


def test_example_trains_and_evaluates_on_one_rank():
    m_ap, world, nb, out = _run(1)
    assert world == 1 and nb == 4 and 0.0 < m_ap <= 1.0
    assert out.count("** Results **") == 1 and "=> Start training" in out


def test_example_trains_and_evaluates_on_two_ranks_from_a_plain_command():
    m_ap, world, nb, out = _run(2, {"IEEE_DIST_BACKEND": "gloo", "IEEE_FORCE_DEVICE": "0"})
    assert world == 2 and nb == 4 and 0.0 < m_ap <= 1.0
    # rank 0 alone reports (rank 1's stdout goes to the launcher's stderr); the queries were sharded over the ranks
    assert out.count("** Results **") == 1 and "queries sharded over 2 ranks" in out
