"""GPU: the measurement legs bench.py adds for BASELINE config 5 (Market1501-multimodal: 750 identities, 32 triples per GPU, the
reference's ablation sweep -- models/ieee3modalPart.py:312-314, engine/image/softmax.py:81-132) and its command-line form."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_legs_return_finite_lines_and_3m_off_uses_the_softmax_engine():
    sys.path.insert(0, ROOT)
    import bench
    out = bench.config5_legs(torch.device("cuda", 0), bench.PEAK_BF16_TFLOPS, steps=2, warmup=1)
    assert set(out["legs"]) == {"full", "noatt", "nocim", "norem", "3m_off"} and "750 classes" in out["workload"]
    for name, leg in out["legs"].items():
        assert "error" not in leg, (name, leg)
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and leg["loss_last_step"] == leg["loss_last_step"]
        assert 0 < leg["whole_step_frac_of_peak"] < 1
        assert leg["engine"] == ("MultiModalImageSoftmaxEngine" if name == "3m_off" else "Image3MEngine")


def test_bench_command_line_runs_one_ablation_leg():
    """`python bench.py --classes 750 --batch 32 --ablation nocim` (what the 4-GPU sweep runs per leg, here on one GPU): ONE
    JSON line whose config names the leg"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--classes", "750", "--batch", "32", "--ablation", "nocim",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-distmat", "--no-fp32", "--no-roofline-pass"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=570)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["config"]["classes"] == 750 and d["config"]["ablation"] == "nocim" and d["config"]["global_batch"] == 32
    assert "CIM off" in d["config"]["workload"] and "config5" not in d and d["n_gpus"] == 1
