#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): 3-modal images/s of the IEEE3modalPart TRAIN STEP
(forward + CE x18 + 3M + backward + [RCCL grad all-reduce] + SGD-nesterov) on synthetic RGBNT201-shaped
batches, config "Single-GPU MI355X: RGBNT201 256x128 batch=64, full CIM+REM+3M, bf16" per GPU
(weak scaling: 64 triples per rank), plus the evaluator's query x gallery distmat GFLOP/s.

  python bench.py --gpus N --steps K --warmup W
  (N>1: that plain command starts its own N ranks, one process per GPU (ieee_amd.dist.launch); equally
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one Image3MEngine.forward_backward over one resident batch (inputs already in HBM).
Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (conv_gather_kernel: implicit
GEMM forward+dgrad): algorithmic FLOPs of its launches / their summed durations, measured with HIP
events on the launch stream in a second pass over the same K steps (the event pairs would perturb the
timed region).  `cpu_baseline` is the oracle's CPU restatement of the same step on the host cores."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# before the HIP runtime starts: ieee_amd/__init__.py picks the hardware queues per stream priority (2 on one GPU, 1 in a
# multi-process job) unless the caller exported a value; the ranks this file starts itself (--gpus N from a plain `python`)
# get the caller's value or 1
_USER_QUEUES = os.environ.get("GPU_MAX_HW_QUEUES")
import ieee_amd  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
TRAIN_GFLOP_PER_TRIPLE = 92.24
# Algorithmic HBM floor of one train step (LABNOTES.md §5, SURVEY.md §8d's counting): every conv output written once and
# read once in the forward (26.47 M elements per triple, bf16), the same bytes again for the dgrad pass (dY read, dX
# written) and for the wgrad operands (dY and X read); SGD-nesterov reads p, g, momentum and writes p, momentum (20 B per
# parameter); the packed bf16 operands (forward + dgrad form) are written and read once each.
CONV_OUT_ELEMS_PER_TRIPLE = 26.47e6
N_PARAMS = 109499337
N_CONV_PARAMS = 95714496          # backbones + convOne + convAvgRest


def step_floor_bytes(B, esz=2):
    act = 3 * (CONV_OUT_ELEMS_PER_TRIPLE * esz * 2) * B
    return act + 20.0 * N_PARAMS + 2 * 2 * esz * N_CONV_PARAMS


class _FakeDM(object):
    def __init__(self, num_classes):
        self.num_train_pids = num_classes
        self.train_loader = []
        self.test_loader = {}
        self.sources = ["synthetic"]


def make_batch(B, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    imgs = [torch.randn(B, 3, 256, 128, generator=g).to(device) for _ in range(3)]
    pids = (torch.arange(B) // 4).to(device)
    return {"img": imgs, "pid": pids, "camid": torch.zeros(B, dtype=torch.long), "impath": "",
            "timeid": torch.zeros(B, dtype=torch.long)}


def host_cores():
    """(physical cores, logical CPUs) of the host: distinct (package, core) pairs of /proc/cpuinfo"""
    logical = os.cpu_count() or 1
    try:
        pairs, pkg = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((pkg, line.split(":")[1].strip()))
        return (len(pairs) or None), logical
    except Exception:
        return None, logical


def committed_traffic(key, field="hbm_bytes_per_launch"):
    """HBM bytes per launch from the newest committed PMC summary (profiles/rNN_pmc_*.json); these come from separate
    rocprofv3 --pmc passes (FETCH_SIZE doubled per the guide's gfx950 note, + WRITE_SIZE), NOT from this run"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True):
        try:
            d = json.load(open(path))
            if key in d and field in d[key]:
                return d[key][field], "%s (separate rocprofv3 --pmc passes%s; not measured in this run)" % (
                    os.path.relpath(path, ROOT), ", build " + d["_build"] if "_build" in d else "")
        except Exception:
            continue
    return None, None


FWD_DGRAD_KERNELS = ("conv_gather_kernel", "conv3x3_patch_kernel", "stem_conv_kernel")
WGRAD_KERNELS = ("conv_wgrad_kernel", "conv3x3_wgrad_patch_kernel", "stem_wgrad_kernel", "wgrad_reduce")


def committed_kernel_stats(train_gflop_per_triple):
    """The same two families from the newest committed `rocprofv3 --kernel-trace --stats` summary of a PLAIN run of this
    bench (profiles/rNN_kernel_stats_in_situ.csv + .json naming its command, build and step count: no event passes in the
    trace, every step is a two-stream step): total duration of the family / steps -> TFLOP/s inside the step.  Read from
    the committed file -- the tracer cannot run inside this process -- and labelled as such."""
    import csv
    import glob
    for meta_path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_in_situ.json")), reverse=True):
        try:
            meta = json.load(open(meta_path))
            rows = list(csv.DictReader(open(meta_path[:-5] + ".csv")))
            steps = float(meta["steps_in_trace"])
            def fam(names):
                ns = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in names))
                calls = sum(int(r["Calls"]) for r in rows if any(k in r["Name"] for k in names))
                return ns / steps * 1e-6, calls / steps
            g_ms, g_calls = fam(FWD_DGRAD_KERNELS)
            w_ms, w_calls = fam(WGRAD_KERNELS)
            fl = float(meta["fwd_dgrad_flops_per_step"])
            return {"source": os.path.relpath(meta_path[:-5] + ".csv", ROOT), "build": meta.get("build"), "command": meta.get("command"),
                    "steps_in_trace": steps, "fwd_dgrad_ms_per_step": g_ms, "fwd_dgrad_launches_per_step": g_calls,
                    "fwd_dgrad_TFLOPs": fl / (g_ms * 1e-3) / 1e12, "fwd_dgrad_frac": fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                    "wgrad_incl_reduce_ms_per_step": w_ms,
                    "wgrad_incl_reduce_TFLOPs": float(meta["wgrad_flops_per_step"]) / (w_ms * 1e-3) / 1e12,
                    "wgrad_incl_reduce_frac": float(meta["wgrad_flops_per_step"]) / (w_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
        except Exception:
            continue
    return None


def committed_layer_ceiling(batch):
    """What the forward + dgrad launches of THIS layer mix could reach if every launch sat on its own roof: per row of the
    newest committed per-layer table (profiles/rNN_layers.csv: FLOPs and algorithmic bytes per launch, B = 64) the time
    max(FLOPs / MFMA peak, algorithmic bytes / HBM peak), summed.  layer1 / layer2 are HBM-bound at 50-100 FLOP per byte, so
    the family's ceiling is well below the MFMA peak that `frac` is quoted against; a second figure uses what the guide's
    best kernels reach (1.27 PFLOP/s for a GEMM, 5.5 TB/s for a stream) instead of the nominal peaks."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_layers.csv")), reverse=True):
        try:
            rows = [r for r in csv.DictReader(open(path)) if r["kind"] in ("fwd", "dgrad")]
            fl = sum(int(r["launches_per_step"]) * float(r["gflop"]) * 1e9 for r in rows)
            def at(pf, bw):
                return sum(int(r["launches_per_step"]) * max(float(r["gflop"]) * 1e9 / pf, float(r["algorithmic_bytes"]) / bw) for r in rows)
            t_nom, t_prac = at(PEAK_BF16_TFLOPS * 1e12, PEAK_HBM_GBS * 1e9), at(1.27e15, 5.5e12)
            if batch != 64 or fl <= 0:
                return None
            return {"source": os.path.relpath(path, ROOT), "nominal": {"ms_per_step": t_nom * 1e3, "TFLOPs": fl / t_nom / 1e12,
                                                                        "frac_of_mfma_peak": fl / t_nom / 1e12 / PEAK_BF16_TFLOPS},
                    "practical": {"ms_per_step": t_prac * 1e3, "TFLOPs": fl / t_prac / 1e12, "gemm_TFLOPs": 1270.0, "stream_GBps": 5500.0},
                    "what": "sum over the 109 forward + dgrad launches of max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s); practical: "
                            "against 1.27 PFLOP/s and 5.5 TB/s, what the guide's best GEMM and streaming kernels reach"}
        except Exception:
            continue
    return None


def cpu_baseline_train(seconds_budget=14.0):
    """oracle (stock torch CPU ops arranged like the reference) on a bounded sample: B=8 steps.  Two legs: 16 threads (the
    headline `value`: the fastest setting on these small convs) and one thread per PHYSICAL core of the host (`all_cores`),
    each bounded by `seconds_budget`, so that the figure is not an artefact of the thread count chosen here."""
    from ieee_amd import detgen
    from ieee_amd._spec import state_spec
    from oracle import model as om
    C = 171
    shapes = {k: s for k, s, _ in state_spec(C)}
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in detgen.generate_state(shapes, 0, rem_param=0.0).items()}
    B = 8
    xs = [torch.from_numpy(x) for x in detgen.generate_images(B, 0)]
    pids = torch.arange(B) // 4
    # torch's default (one thread per hardware thread, 128 on the GPU box) oversubscribes these small convs
    # and is ~5x slower than 16 threads; the baseline uses 16 and says so
    default_threads = torch.get_num_threads()
    phys, logical = host_cores()

    def leg(threads):
        torch.set_num_threads(threads)
        om.train_step(sd, xs, pids, C)                       # warm-up
        t0, n = time.time(), 0
        while n < 2 or (time.time() - t0 < seconds_budget and n < 6):
            om.train_step(sd, xs, pids, C)
            n += 1
        return (time.time() - t0) / n, n
    threads = min(16, default_threads)
    dt, n = leg(threads)
    out = {"value": B / dt, "unit": "3-modal images/s", "cores": threads, "threads": threads, "kind": "port",
           "host_physical_cores": phys, "host_logical_cpus": logical,
           "why_16_threads": "the step's convs at batch 8 are small: beyond ~16 threads torch's CPU kernels lose more to "
                             "synchronisation than they gain (see all_cores)",
           "sample": "%d oracle train steps at batch %d (fp32, torch CPU ops, %d threads of a host with %s physical cores / "
                     "%d logical CPUs), %.2f s/step" % (n, B, threads, phys, logical, dt),
           "note": "a SMALL-BATCH figure (8 triples per step) beside a 64-triple GPU step: bounded sample, baseline only.  torch's "
                   "CPU convs at batch 8 do not scale past ~16 threads (all_cores), which says nothing about the host's speed"}
    wide = phys or default_threads
    if wide and wide > threads:
        try:
            dt2, n2 = leg(int(wide))
            out["all_cores"] = {"value": B / dt2, "unit": "3-modal images/s", "cores": int(wide), "steps": n2,
                                "sample": "%d oracle train steps at batch %d on %d threads (one per physical core), %.2f s/step" % (n2, B, wide, dt2)}
        except Exception as e:
            out["all_cores"] = {"error": str(e)}
    torch.set_num_threads(default_threads)
    return out


def dp_path_leg(engine, batch, model, rounds=3, steps=12):
    """staged data-parallel step (IEEE_FORCE_DP_PATH=1, 1-rank "nccl" = RCCL group) against the plain step, interleaved"""
    import socket
    import torch.distributed as dist
    made = False
    try:
        if not dist.is_initialized():
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
            made = True

        def run(forced, n, overlap=True):
            os.environ["IEEE_FORCE_DP_PATH"] = "1" if forced else "0"
            engine.dp_overlap = overlap
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(n):
                engine.forward_backward(batch)
            torch.cuda.synchronize()
            return (time.time() - t0) / n * 1e3
        # communicator set-up is not a step: RCCL registers the gradient buffer at its first collectives (the first ~20
        # staged steps run 1.3-1.6x slower)
        for ranges in model.grad_part_ranges():
            for a, b in ranges:
                dist.all_reduce(model._flat_grads[a:b], op=dist.ReduceOp.SUM)
        run(True, 25)
        run(True, 6, overlap=False)
        plain, staged, simple = [], [], []
        for _ in range(rounds):
            plain.append(run(False, steps))
            staged.append(run(True, steps))
            simple.append(run(True, steps, overlap=False))
        p, st, sm = (sorted(v)[len(v) // 2] for v in (plain, staged, simple))
        return {"plain_ms_per_step": p, "staged_ms_per_step": st, "dp_path_overhead_ms": st - p,
                "unoverlapped_dp_ms_per_step": sm, "backend": dist.get_backend(),
                "ranks": dist.get_world_size(), "rounds": rounds, "steps_per_round": steps,
                "what": "the N > 1 step (5 backward parts, 13 gradient slices all-reduced from a communication stream, optimizer "
                        "slices behind them) run over a 1-rank RCCL group on this GPU, interleaved with the plain step: the fixed "
                        "per-rank cost of the data-parallel code path without any wire time; unoverlapped = the fallback form "
                        "(whole backward, one all-reduce pass on the compute stream, one optimizer step; IEEE_DP_OVERLAP=0)"}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        os.environ["IEEE_FORCE_DP_PATH"] = "0"
        engine.dp_overlap = None
        if made:
            try:
                dist.destroy_process_group()
            except Exception:
                pass


def bench_distmat(device):
    """config C4: 10k x 100k x 768 squared-Euclidean distmat (fp32, exact) + CMC/mAP on the device"""
    from ieee_amd.metrics import compute_distance_matrix, evaluate_rank
    g = torch.Generator(device="cpu").manual_seed(1)
    Q, G, D = 10000, 100000, 768
    qf = torch.randn(Q, D, generator=g).abs().to(device)
    gf = torch.randn(G, D, generator=g).abs().to(device)
    rs = np.random.RandomState(1)
    qp, gp = rs.randint(0, 1000, Q), rs.randint(0, 1000, G)
    qc, gc = rs.randint(0, 4, Q), rs.randint(0, 4, G)
    out = {}
    for name, a, b in (("fp32", qf, gf), ("bf16", qf.bfloat16(), gf.bfloat16())):
        compute_distance_matrix(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            dm = compute_distance_matrix(a, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": ms, "GFLOP/s": 2.0 * Q * G * D / ms / 1e6}
    # fp32 rows as exact 16-bit pieces on the bf16 / fp16 matrix cores (6 resp. 3 piece products: fp32-grade accuracy)
    for name, prec, terms in (("split_bf16x3", "bf16x3", 6), ("split_f16x2", "f16x2", 3)):
        compute_distance_matrix(qf, gf, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dm = compute_distance_matrix(qf, gf, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[name] = {"ms": ms, "GFLOP/s": 2.0 * Q * G * D / ms / 1e6, "mfma_TFLOP/s": terms * 2.0 * Q * G * D / ms / 1e9,
                     "mfma_frac_of_16bit_peak": terms * 2.0 * Q * G * D / ms / 1e9 / PEAK_BF16_TFLOPS,
                     "note": "includes the piece-splitting pre-pass; GFLOP/s counts the 2*Q*G*D of the fp32 problem"}
    del dm
    # the model's real descriptor width (2304 = 3 x 768, ieee3modalPart.py:502), fp32, the full 10 k x 100 k problem
    g2 = torch.Generator(device="cpu").manual_seed(2)
    q3, g3 = torch.randn(Q, 2304, generator=g2).abs().to(device), torch.randn(G, 2304, generator=g2).abs().to(device)
    d3 = compute_distance_matrix(q3, g3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        d3 = compute_distance_matrix(q3, g3)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out["fp32_d2304"] = {"ms": ms, "GFLOP/s": 2.0 * Q * G * 2304 / ms / 1e6, "workload": "%d x %d x 2304" % (Q, G),
                         "frac_of_fp32_mfma_peak": 2.0 * Q * G * 2304 / ms / 1e9 / PEAK_F32_TFLOPS}
    del q3, g3, d3
    dm = compute_distance_matrix(qf, gf)
    evaluate_rank(dm, qp, gp, qc, gc)
    torch.cuda.synchronize()
    t0 = time.time()
    cmc, m_ap = evaluate_rank(dm, qp, gp, qc, gc)
    torch.cuda.synchronize()
    out["rank_ms"] = (time.time() - t0) * 1e3          # wall time of evaluate_rank (id upload + kernels + read-back)
    # the ranking kernels alone, HIP events on their stream, ids already on the device: HBM-bound (4 B per pair)
    from ieee_amd import _lib
    lib = _lib.load()
    ids = [torch.from_numpy(a.astype(np.int32)).to(device) for a in (qp, gp, qc, gc)]
    ap_d = torch.empty(Q, dtype=torch.float64, device=device)
    first_d = torch.empty(Q, dtype=torch.int32, device=device)
    summ_d = torch.empty(22, dtype=torch.int64, device=device)
    rk_work = torch.empty(lib.ieee_rank_workspace_bytes(G), dtype=torch.uint8, device=device)
    def rank_call():      # what evaluate_rank calls (identity buckets built inside: part of the timed region)
        _lib.check(lib.ieee_rank_market1501_ws(_lib.ptr(dm), dm.stride(0), Q, G, _lib.ptr(ids[0]), _lib.ptr(ids[1]),
                                               _lib.ptr(ids[2]), _lib.ptr(ids[3]), 20, _lib.ptr(ap_d), _lib.ptr(first_d),
                                               _lib.ptr(summ_d), _lib.ptr(rk_work), rk_work.numel(), _lib.stream()))
    rank_call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        rank_call()
    e1.record()
    torch.cuda.synchronize()
    rk_ms = e0.elapsed_time(e1) / 5
    out["rank_kernels_ms"] = rk_ms
    rk_traffic, rk_src = committed_traffic("rank_query_fast")
    out["roofline_rank"] = {"bound": "hbm", "achieved": 4.0 * Q * G / rk_ms / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": 4.0 * Q * G / rk_ms / 1e6 / PEAK_HBM_GBS, "traffic": rk_traffic, "traffic_source": rk_src}
    out["workload"] = "10000 x 100000 x 768 (BASELINE config 4)"
    out["roofline_fp32"] = {"bound": "mfma", "achieved": out["fp32"]["GFLOP/s"] / 1e3, "peak": PEAK_F32_TFLOPS,
                            "unit": "TFLOP/s", "frac": out["fp32"]["GFLOP/s"] / 1e3 / PEAK_F32_TFLOPS}
    # CPU baselines on bounded samples: oracle sgemm-form distmat, reference's own Cython evaluator
    try:
        from oracle import evaluator as ev
        q2, g2 = qf[:2000].cpu().numpy(), gf[:20000].cpu().numpy()
        t0 = time.time()
        d2 = ev.sqeuclid_np(q2, g2)
        dt = time.time() - t0
        out["cpu_distmat"] = {"GFLOP/s": 2.0 * 2000 * 20000 * D / dt / 1e9, "kind": "port", "sample": "2000 x 20000 x 768 numpy"}
        t0 = time.time()
        ev.rank_market1501_c(d2[:300, :3000].copy(), qp[:300], gp[:3000], qc[:300], gc[:3000])
        out["cpu_rank_ms_300x3000"] = (time.time() - t0) * 1e3
    except Exception as e:      # the baseline is informative only
        out["cpu_distmat"] = {"error": str(e)}
    return out


METRIC = "3-modal images/s (train fwd+bwd)"
# BASELINE config 5's sweep (reference README.md:44-101): model flags of models/ieee3modalPart.py:312-314, and the CE-only
# engine of engine/image/softmax.py:81-132 for "3M off"
ABLATIONS = {"full": {}, "noatt": {"attention": False}, "nocim": {"interaction": False}, "norem": {"using_REM": False},
             "3m_off": {}}


def make_engine(C, ablation, cdt, device):
    """(model, engine) of one leg: `ablation` picks the model flags and, for 3m_off, the CE-only engine"""
    from ieee_amd.engine import Image3MEngine, MultiModalImageSoftmaxEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    softmax = ablation == "3m_off"
    model = build_model("ieee3modalPart", num_classes=C, loss="softmax" if softmax else "margin", pretrained=False, use_gpu=True,
                        compute_dtype=cdt, device=device, **ABLATIONS[ablation])
    opt = build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9)
    if softmax:
        engine = MultiModalImageSoftmaxEngine(_FakeDM(C), model, opt, use_gpu=True, label_smooth=True)
    else:
        engine = Image3MEngine(_FakeDM(C), model, opt, margin=1, weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
    return model, engine


def workload_text(B, C, ablation):
    what = {"full": "full CIM+REM+3M", "noatt": "attention off (CIM without the channel attention), REM+3M",
            "nocim": "CIM off (interaction=False), REM+3M", "norem": "REM off, CIM+3M",
            "3m_off": "3M off (CE-only MultiModalImageSoftmaxEngine), CIM+REM"}[ablation]
    data = "RGBNT201-shaped" if C == 171 else ("Market1501-multimodal-shaped" if C == 750 else "synthetic")
    return "IEEE3modalPart train step, %s 256x128 triples, batch %d per GPU, %d classes, %s, SGD-nesterov" % (data, B, C, what)


def config5_legs(device, peak_tflops, steps=10, warmup=3, B=32, C=750):
    """BASELINE config 5 on ONE GPU: Market1501-multimodal's 750 identities, 32 triples per GPU, the reference's ablation sweep
    {full, attention off, CIM off, REM off, 3M off}, bf16 -- five short legs (fresh model + engine each), images/s and
    the whole step's fraction of the bf16 MFMA peak (the full model's 92.24 GFLOP per triple for every leg: the ablated
    legs do a little less work, so their fraction is an upper bound)."""
    out = {"workload": "IEEE3modalPart train step, 256x128 triples, batch %d per GPU, %d classes, bf16; one leg per ablation" % (B, C),
           "steps": steps, "warmup": warmup, "legs": {}}
    batch = make_batch(B, seed=5, device=device)
    for name in ("full", "noatt", "nocim", "norem", "3m_off"):
        try:
            model, engine = make_engine(C, name, torch.bfloat16, device)
            engine.resident_batch = True
            engine.defer_summary = True
            model.train()
            for _ in range(warmup):
                engine.forward_backward(batch)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(steps):
                summary = engine.forward_backward(batch)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / steps
            loss = float(summary["loss_all" if name == "3m_off" else "loss"])
            out["legs"][name] = {"value": B / dt, "unit": "3-modal images/s", "ms_per_step": dt * 1e3, "loss_last_step": loss,
                                 "engine": type(engine).__name__,
                                 "whole_step_frac_of_peak": B / dt * TRAIN_GFLOP_PER_TRIPLE * 1e9 / (peak_tflops * 1e12)}
            del engine, model
        except Exception as e:      # informative leg: never costs the headline line
            out["legs"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 50 timed steps after 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="triples per GPU (BASELINE config 2/3: 64)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-distmat", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the short fp32 parity-mode throughput leg")
    ap.add_argument("--no-roofline-pass", action="store_true", help="plain timed steps only (the command rocprofv3 traces for "
                    "profiles/rNN_kernel_stats_in_situ.csv: every step in the trace is a two-stream step)")
    ap.add_argument("--no-dp-path", action="store_true", help="skip the staged data-parallel step over a 1-rank RCCL group (N = 1)")
    ap.add_argument("--no-loader", action="store_true", help="skip the input-pipeline leg (JPEG tree -> loader -> train step)")
    ap.add_argument("--loader-workers", default="8,16,32", help="worker counts of the input-pipeline leg")
    ap.add_argument("--classes", type=int, default=171, help="identities = classifier width (RGBNT201: 171; Market1501-multimodal, "
                    "BASELINE config 5: 750)")
    ap.add_argument("--ablation", default="full", choices=sorted(ABLATIONS), help="BASELINE config 5's sweep: full, noatt "
                    "(attention=False), nocim (interaction=False), norem (using_REM=False), 3m_off (the CE-only "
                    "MultiModalImageSoftmaxEngine); reference models/ieee3modalPart.py:312-314, engine/image/softmax.py:81-132")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 object of the default N = 1 line (5 short legs "
                    "at 750 classes, 32 triples)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="N > 1 from a plain `python`: seconds the self-started "
                    "ranks get before they are ended and an error line is printed (exit code 124)")
    args = ap.parse_args()

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1") or 1) == 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher -- N child interpreters, one rank per GPU, with
        # the environment torchrun would give them (ieee_amd.dist.launch).  It never touches the GPU and nothing is exec'd;
        # rank 0 inherits stdout (the JSON line), the other ranks' stdout goes to stderr; exit code = the first failure.
        # A rank that hangs (its first RCCL collective has never met real hardware) cannot hang the job silently: after
        # --launch-timeout seconds, or when a rank fails, every rank is ended and ONE JSON error line goes to stdout.
        from ieee_amd import dist as ddp
        report = {}
        rc = ddp.launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, queues=_USER_QUEUES,
                        timeout=args.launch_timeout, report=report, capture_stderr=True)
        if rc != 0:
            what = ("timeout: the ranks were still running after %.0f s and were ended" % args.launch_timeout) if report.get("timed_out") \
                else ("the launcher received signal %d" % (rc - 128) if report.get("rank") is None and rc > 128
                      else "rank %s exited with code %d" % (report.get("rank"), rc))
            print(json.dumps({"metric": METRIC, "value": None, "unit": "3-modal images/s", "n_gpus": args.gpus,
                              "error": what, "rank": report.get("rank"), "exit_code": rc,
                              "elapsed_s": report.get("elapsed_s"), "stderr_tail": report.get("stderr_tail", "")}), flush=True)
        sys.exit(rc)

    # Only the JSON line may reach the caller's stdout: RCCL prints a version banner (through C stdio, flushed at exit) when
    # a communicator is created, and other libraries chat too.  From here on file descriptor 1 IS stderr; the line is written
    # to the saved descriptor at the very end.
    if os.environ.get("IEEE_BENCH_TEST_HANG_RANK") in (os.environ.get("RANK", "0"), "all"):
        time.sleep(1e6)          # test hook (tests/test_dist_cpu.py): this rank never arrives -- the launcher's timeout must end the job
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from ieee_amd import _lib, dist as ddp
    world, rank, local = ddp.init_from_env()
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under torchrun with N ranks" % (args.gpus, world)
    _lib.require_gpu()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    C = args.classes
    torch.manual_seed(0)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model, engine = make_engine(C, args.ablation, cdt, device)
    headline = C == 171 and args.ablation == "full"     # BASELINE config 2 / 3: what the committed profiles describe
    engine.dp_presharded = True          # weak scaling: every rank generates its own 64 triples (identity-aligned)
    engine.dp_total_rows = args.batch * world    # ... so the global batch is known without asking the other ranks
    engine.resident_batch = True         # the same batch object every step: its 3M chunk check runs once
    # the loss summary of a step is read back when somebody looks at it (here: after the timed loop), not inside the
    # step: the host enqueues step k+1 while step k runs, as a training loop that prints every print_freq batches does
    engine.defer_summary = os.environ.get("IEEE_DEFER_SUMMARY", "1") != "0"
    model.train()
    B = args.batch
    batch = make_batch(B, seed=rank, device=device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    dp_calibration = None
    if world > 1:
        # communicator set-up is not a training step: RCCL builds its channels and registers the gradient buffer at the
        # first collectives over it (measured with a 1-rank group: the first ~20 staged steps run 1.3-1.6x slower), so
        # touch every slice of the flat gradient once before the warm-up steps
        for ranges in model.grad_part_ranges():
            for a, b in ranges:
                torch.distributed.all_reduce(model._flat_grads[a:b], op=torch.distributed.ReduceOp.SUM)
        model._flat_grads.zero_()
        barrier()
        if os.environ.get("IEEE_DP_OVERLAP") is None:
            # which form of the data-parallel step runs: the overlapped one (5 backward parts, gradient slices all-reduced
            # from a communication stream while the next part computes, optimizer slices behind them) unless the plain one
            # (whole backward, one all-reduce pass, one update) is clearly faster HERE -- how the runtime maps the step's
            # streams and RCCL's onto hardware queues decides whether the overlap materialises (LABNOTES.md, "Round 4"), and a
            # single-GPU box cannot tell.  Timed on every rank, decided on the slowest rank's figures.
            def cal_steps(n):
                barrier()
                t0 = time.time()
                for _ in range(n):
                    engine.forward_backward(batch)
                barrier()
                return (time.time() - t0) / n * 1e3
            cal, cal_err = [], None
            for ov in (True, False):
                engine.dp_overlap = ov
                try:
                    cal_steps(6)
                    cal.append(cal_steps(8))
                except Exception as e:
                    # the overlapped form (collectives from a communication stream beside the backward) threw: the plain
                    # form is the fallback.  (A throw of the plain form has no fallback: it propagates.)
                    if not ov:
                        raise
                    cal_err = "%s: %s" % (type(e).__name__, e)
                    cal.append(float("inf"))
            t = torch.tensor(cal, dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            engine.dp_overlap = bool(float(t[0]) <= 1.02 * float(t[1]))
            dp_calibration = {"overlapped_ms_per_step": float(t[0]) if float(t[0]) != float("inf") else None,
                              "unoverlapped_ms_per_step": float(t[1]),
                              "used": "overlapped" if engine.dp_overlap else "unoverlapped", "steps_each": 8}
            if cal_err is not None:
                dp_calibration["overlapped_error"] = cal_err
    if os.environ.get("IEEE_BENCH_HIPRIO") == "1":       # experiment: the step's launch stream as a high-priority stream
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=-1))
    for _ in range(args.warmup):
        summary = engine.forward_backward(batch)
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        summary = engine.forward_backward(batch)
    barrier()
    dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    value = args.steps * B * world / dt

    # ---- roofline passes: the same steps with a HIP event pair around every conv launch.  IN SITU (mode 2): the executor's
    # streams stay on -- forward / dgrad pairs on the launch stream, weight-gradient pairs on the side stream -- so a pair
    # times its launch beside the other stream's kernels: what the launch takes inside the step that `value` measures, and
    # what `rocprofv3 --kernel-trace --stats` of the same command reports (profiles/).  SERIALIZED (mode 1): everything
    # on one ordered stream, every launch alone on the machine.
    import ctypes
    net = model.native_net(B, 256, 128)
    lib = _lib.load()

    def event_pass(mode):
        _lib.check(lib.ieee_net_profile(net.handle, mode, None))
        for _ in range(args.steps):
            engine.forward_backward(batch)
        o = (ctypes.c_double * 6)()
        _lib.check(lib.ieee_net_profile(net.handle, 0, o))
        return list(o)
    if args.no_roofline_pass:
        g_ms = g_fl = g_n = w_ms = w_fl = w_n = sg_ms = sg_fl = sg_n = sw_ms = sw_fl = sw_n = 0.0
    else:
        g_ms, g_fl, g_n, w_ms, w_fl, w_n = event_pass(2)
        sg_ms, sg_fl, sg_n, sw_ms, sw_fl, sw_n = event_pass(1)
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
    ser = sg_fl / (sg_ms * 1e-3) / 1e12 if sg_ms > 0 else 0.0
    traffic = traffic_src = None
    if args.dtype == "bf16" and B == 64 and headline:
        traffic, traffic_src = committed_traffic("conv_fwd_dgrad")          # gather + LDS-patch + direct-stem launches (round 3)
        if traffic is None:
            traffic, traffic_src = committed_traffic("conv_gather")
    stats_line = committed_kernel_stats(TRAIN_GFLOP_PER_TRIPLE) if args.dtype == "bf16" and B == 64 and headline else None
    ceiling = committed_layer_ceiling(B) if args.dtype == "bf16" else None
    roofline = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "measured": "in situ: every launch of the family carries a HIP event pair as its own start / stop signals "
                            "(hipExtLaunchKernelGGL) inside the two-stream step; nothing is added to the queues",
                "in_situ_frac": ach / peak, "serialized_achieved": ser, "serialized_frac": ser / peak,
                "serialized_avg_launch_us": sg_ms * 1e3 / max(sg_n, 1),
                "in_situ_from_committed_rocprof_stats": stats_line,
                "layer_mix_ceiling": ceiling,
                "frac_of_layer_mix_ceiling": ({"in_situ": ach / ceiling["nominal"]["TFLOPs"], "serialized": ser / ceiling["nominal"]["TFLOPs"],
                                               "in_situ_vs_practical": ach / ceiling["practical"]["TFLOPs"],
                                               "serialized_vs_practical": ser / ceiling["practical"]["TFLOPs"]} if ceiling else None),
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "conv forward + dgrad launches: conv_gather_kernel (implicit GEMM), conv3x3_patch_kernel (3x3 from an "
                          "LDS-resident patch), stem_conv_kernel",
                "launches": int(g_n), "avg_launch_us": g_ms * 1e3 / max(g_n, 1),
                "flops_per_launch": g_fl / max(g_n, 1), "fwd_dgrad_flops_per_step": g_fl / args.steps,
                "wgrad_flops_per_step": w_fl / args.steps,
                "wgrad": {"achieved": (w_fl / (w_ms * 1e-3) / 1e12) if w_ms > 0 else 0.0, "launches": int(w_n),
                          "avg_launch_us": w_ms * 1e3 / max(w_n, 1), "frac": (w_fl / (w_ms * 1e-3) / 1e12 / peak) if w_ms > 0 else 0.0,
                          "serialized_achieved": (sw_fl / (sw_ms * 1e-3) / 1e12) if sw_ms > 0 else 0.0,
                          "note": "weight-gradient kernels incl. their slab reductions; in situ = on the side stream, beside the "
                                  "launch stream's kernels"},
                # own start / stop signals bracket [the command processor starts the dispatch -> its completion signal]: a few
                # microseconds of dispatch latency per launch that rocprofv3's kernel begin / end timestamps do not contain.
                # That constant per launch is the whole gap between `in_situ_frac` and the committed rocprofv3 figure.
                "in_situ_minus_rocprof_us_per_launch": ((g_ms * 1e3 / max(g_n, 1)) - stats_line["fwd_dgrad_ms_per_step"] * 1e3 /
                                                        max(stats_line["fwd_dgrad_launches_per_step"], 1)) if (stats_line and g_n) else None,
                "conv_ms_per_step": (g_ms + w_ms) / args.steps,
                "whole_step_frac_of_peak": value / world * TRAIN_GFLOP_PER_TRIPLE * 1e9 / (peak * 1e12)}

    # whole-step HBM traffic: PMC bytes per step (committed summary of separate rocprofv3 --pmc passes) against the
    # algorithmic floor, and the rate / fraction of the 8 TB/s peak they mean at THIS run's step time
    step_hbm = None
    if args.dtype == "bf16" and B == 64 and headline:
        pmc_bytes, pmc_src = committed_traffic("_step", "hbm_bytes_per_step")
        floor = step_floor_bytes(B)
        step_s = dt / args.steps
        step_hbm = {"pmc_bytes_per_step": pmc_bytes, "pmc_source": pmc_src, "algorithmic_floor_bytes": floor,
                    "ratio_to_floor": (pmc_bytes / floor) if pmc_bytes else None,
                    "achieved_TBps": (pmc_bytes / step_s / 1e12) if pmc_bytes else None,
                    "frac_of_hbm_peak": (pmc_bytes / step_s / 1e9 / PEAK_HBM_GBS) if pmc_bytes else None,
                    "floor_TBps_at_this_step_time": floor / step_s / 1e12,
                    "floor": "3 passes x (conv outputs written + read once, bf16) + SGD 20 B/param + packed operands "
                             "written + read once (LABNOTES.md section 5)"}

    # N = 1: the fixed cost of the N > 1 code path.  The SAME engine runs the staged data-parallel step -- backward in 5
    # parts, every part's gradient slices all-reduced over a 1-rank RCCL group from the communication stream, the
    # optimizer slices behind them (engine.py: _fused_step, `staged`) -- interleaved with the plain step.
    dp_path = None
    if world == 1 and not args.no_dp_path and args.dtype == "bf16" and headline:
        dp_path = dp_path_leg(engine, batch, model)

    # N > 1: what RCCL saw -- rank count and the time of each backward part's gradient all-reduce (a short extra leg
    # with event pairs on the communication stream; not part of the timed region)
    rccl = None
    if world > 1:
        used_overlap = getattr(engine, "dp_overlap", None)     # (None: IEEE_DP_OVERLAP decided, no calibration ran)
        engine.dp_overlap = True             # the per-part event pairs exist in the overlapped form
        engine.time_collectives = []
        for _ in range(min(args.steps, 10)):
            engine.forward_backward(batch)
        torch.cuda.synchronize()
        engine.dp_overlap = used_overlap
        parts = {}
        for part, e0, e1, nbytes in engine.time_collectives:
            parts.setdefault(part, []).append((e0.elapsed_time(e1), nbytes))
        engine.time_collectives = None
        rccl = {"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                "allreduce_ms_per_part": {str(k): sum(v[0] for v in vs) / len(vs) for k, vs in sorted(parts.items())},
                "allreduce_bytes_per_part": {str(k): vs[0][1] for k, vs in sorted(parts.items())},
                "note": "per backward part (0 head+CIM, 1 layer4, 2 layer3, 3 layer2, 4 layer1+stem), measured on rank 0's "
                        "communication stream; the slices overlap the next part's backward"}

    line = {
        "metric": METRIC,
        "value": value,
        "unit": "3-modal images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": workload_text(B, C, args.ablation),
                   "global_batch": B * world, "parallelism": "dp%d" % world, "classes": C, "ablation": args.ablation,
                   "engine": type(engine).__name__,
                   "loss_last_step": float(summary["loss_all" if args.ablation == "3m_off" else "loss"]),
                   "summary_readback": "on first look (engine.defer_summary)" if engine.defer_summary else "inside every step"},
        "roofline": roofline,
    }
    line["launch"] = {"how": "self-spawned" if os.environ.get("IEEE_LAUNCHED_BY") == "ieee_amd.dist.launch" else
                      ("torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else ("single process" if world == 1 else "external")),
                      "hw_queues": ieee_amd.HW_QUEUES, "grad_dtype": engine._grad_dtype() if world > 1 else None}
    if step_hbm is not None:
        line["step_hbm"] = step_hbm
    if rccl is not None:
        line["rccl"] = rccl
    if dp_path is not None:
        line["dp_path"] = dp_path
    if world > 1:
        be = torch.distributed.get_backend()
        line["config"]["dp_step"] = ("overlapped: 5 backward parts, 13 gradient slices all-reduced (%s) from a communication stream, "
                                     "optimizer slices behind them" % be) if getattr(engine, "dp_overlap", None) is not False and os.environ.get("IEEE_DP_OVERLAP", "1") != "0" \
            else "unoverlapped: whole backward, one all-reduce pass (%s), one optimizer step" % be
        if dp_calibration is not None:
            line["dp_calibration"] = dp_calibration
    if rank == 0:
        if world == 1 and not args.no_loader and args.dtype == "bf16" and headline:
            # input pipeline at step rate: JPEG tree -> worker decode -> device transform -> real steps,
            # with THIS engine (the timed one), before the CPU-baseline leg fills the process with OpenMP threads
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import loader_probe
                # three INTERLEAVED rounds per worker count: median and min - max per setting; recommended = the smallest
                # count whose worst round keeps >= 0.97 of the resident-batch rate
                line["loader"] = loader_probe.measure(tuple(int(w) for w in args.loader_workers.split(",")), steps=40, B=B,
                                                      device=device, engine=engine, rounds=3)
            except Exception as e:      # informative leg: never costs the headline line
                line["loader"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_train()
        if world == 1 and (not args.no_distmat or not args.no_fp32):
            del engine, net
            model._nets.clear()
            torch.cuda.empty_cache()
        if world == 1 and args.dtype == "bf16" and headline and not args.no_config5:
            # BASELINE config 5 (Market1501-multimodal, 750 classes, 32 triples per GPU, the ablation sweep) on this GPU; the
            # 4-GPU form is `python bench.py --gpus 4 --classes 750 --batch 32 --ablation <leg>`, one line per leg
            line["config5"] = config5_legs(device, PEAK_BF16_TFLOPS)
        if world == 1 and args.dtype == "bf16" and not args.no_fp32:
            # the fp32 PARITY mode (exact fp32 MFMA end to end: the mode that meets north_star's 1e-3 contract), same
            # workload, a short run: the headline `value` above is the bf16 speed mode
            m32, e32 = make_engine(C, args.ablation, torch.float32, device)
            m32.train()
            for _ in range(2):
                e32.forward_backward(batch)
            torch.cuda.synchronize()
            t1, n32 = time.time(), 6
            for _ in range(n32):
                e32.forward_backward(batch)
            torch.cuda.synchronize()
            d32 = (time.time() - t1) / n32
            line["fp32_parity_mode"] = {"value": B / d32, "unit": "3-modal images/s", "ms_per_step": d32 * 1e3, "steps": n32,
                                        "whole_step_frac_of_fp32_mfma_peak": B / d32 * TRAIN_GFLOP_PER_TRIPLE * 1e9 /
                                        (PEAK_F32_TFLOPS * 1e12)}
            del e32, m32
            torch.cuda.empty_cache()
        if world == 1 and not args.no_distmat:
            line["distmat"] = bench_distmat(device)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    os.close(real_stdout)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
