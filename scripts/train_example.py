#!/usr/bin/env python
"""End-to-end example: the flow of the reference's launcher (scripts/mainMultiModal.py:150-235 -- data manager, build_model,
optimizer, scheduler, engine, engine.run) on ieee_amd, on ONE or SEVERAL GPUs from a plain `python` command.

  python scripts/train_example.py --root /data --epochs 60 --batch 64 --gpus 8
  python scripts/train_example.py --synthetic 16 --epochs 2 --batch 16 --gpus 2      (a generated RGBNT201-layout JPEG tree)

What replaces what: `nn.DataParallel(model).cuda()` (:219-220) -> one process per GPU, started here by ieee_amd.dist.launch
(no torchrun needed); `build_datamanager(cfg)` (:206) -> ieee_amd.data.RGBNT201 + build_loaders (shard-aware: every rank reads
its identity-aligned rows of the reference sampler's global batches); the yacs configuration -> the handful of keyword
arguments RGBNT_ieee_part_margin.yaml sets.  Not a re-implementation of the reference's CLI: an executable INTEGRATION.md."""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_USER_QUEUES = os.environ.get("GPU_MAX_HW_QUEUES")
import ieee_amd  # noqa: E402,F401  (before torch.cuda is touched: it picks GPU_MAX_HW_QUEUES)
from ieee_amd import dist as ddp  # noqa: E402


class DataManager(object):
    """the attributes the engines read from torchreid's ImageDataManager (data/datamanager.py:120-245)"""

    def __init__(self, dataset, train_loader, query_loader, gallery_loader, name, num_instances):
        self.num_train_pids = dataset.num_train_pids
        self.num_train_cams = dataset.num_train_cams
        self.train_loader = train_loader
        self.test_loader = {name: {"query": query_loader, "gallery": gallery_loader}}
        self.sources = [name]
        self.targets = [name]
        self.num_instances = num_instances


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="", help="directory that holds RGBNT201/ (train_171, test with RGB / NI / TI folders)")
    ap.add_argument("--synthetic", type=int, default=0, help="generate a JPEG tree with this many identities instead of --root")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="GLOBAL train batch (identities x 4 instances)")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--eval-freq", type=int, default=-1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--save-dir", default="")
    args = ap.parse_args()

    if args.gpus > 1 and ddp.env_world()[0] == 1:
        # the launcher: N child interpreters, one rank per GPU; this process only waits for them
        sys.exit(ddp.launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, queues=_USER_QUEUES))

    import random
    import numpy as np
    import torch
    from ieee_amd import data as D
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_lr_scheduler, build_optimizer
    world, rank, local = ddp.init_from_env()
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)      # (the reference's seed_torch)

    root = args.root
    if args.synthetic:
        # every rank needs the same files: rank 0 writes them, the others wait at a barrier
        root = os.path.join(tempfile.gettempdir(), "ieee_example_tree_%d" % args.synthetic)
        if rank == 0 and not os.path.isdir(os.path.join(root, "RGBNT201")):
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import loader_probe
            loader_probe.make_tree(root, n_ids=args.synthetic, per_id=8)
        if world > 1:
            torch.distributed.barrier()
    dataset = D.RGBNT201(root=root)
    train, query, gallery = D.build_loaders(dataset, 256, 128, "random_flip", batch_size_train=args.batch, batch_size_test=32,
                                            num_instances=4, workers=args.workers)
    dm = DataManager(dataset, train, query, gallery, "RGBNT201", 4)

    model = build_model("ieee3modalPart", num_classes=dm.num_train_pids, loss="margin", pretrained=False, use_gpu=True,
                        compute_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    optimizer = build_optimizer(model, optim="sgd", lr=args.lr, weight_decay=5e-4, momentum=0.9)
    scheduler = build_lr_scheduler(optimizer, "multi_step", stepsize=[max(1, args.epochs * 2 // 3)], gamma=0.1)
    ranks = [r for r in (1, 5, 10, 20) if r <= len(dataset.gallery)]     # (the reference raises IndexError on a gallery of < 20)
    engine = Image3MEngine(dm, model, optimizer, margin=1, weight_m=1, weight_x=1, scheduler=scheduler, use_gpu=True,
                           label_smooth=True)
    engine.run(save_dir=args.save_dir or os.path.join(tempfile.gettempdir(), "ieee_example_log"), max_epoch=args.epochs,
               eval_freq=args.eval_freq, print_freq=10, ranks=ranks)
    mAP = engine.test(ranks=ranks)           # (Engine.run does not evaluate after the last epoch, like the reference)
    if rank == 0:
        print("FINAL mAP %.4f after %d epoch(s) on %d rank(s), %d train batches per epoch" % (mAP, args.epochs, world, len(train)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
