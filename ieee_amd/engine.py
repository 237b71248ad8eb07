"""Training / evaluation drivers with the reference's call surface (torchreid/engine/engine.py: Engine.run :126-232,
train :234-282, test :287-337, _evaluate :339-441; engine/image/margin.py:62-154 Image3MEngine; engine/image/softmax.py:
11-132 MultiModalImageSoftmaxEngine): same constructor arguments, method names, keyword meaning, printed report and
loss_summary keys, so scripts/mainMultiModal.py-style callers run unchanged.  What is underneath is this package's own:

  * fused step (native model + FusedSGD / FusedAdam): forward, 18-head cross entropy, 3M, backward, the RCCL gradient
    all-reduce and the optimizer update are each one native call; no autograd graph and ONE small device->host read
    per step for the logging dict (the reference synchronises >= 27 times per step).
  * generic step (any other torch.optim optimizer): autograd over the same native forward / backward.
  * data parallel (one process per GPU, ieee_amd/dist.py): the step shards the batch on identity boundaries, the
    replicas are synchronised once before the first step, evaluation shards the feature extraction (one all-gather of
    the descriptors) and the ranking (22 numbers all-reduced) under rank 0's running statistics.
  * evaluation keeps descriptors, the distance matrix and the ranking on the device (the reference copies every
    feature batch to the host and runs both on the CPU, engine.py:368-417)."""
from __future__ import absolute_import, division, print_function

import datetime
import os
import os.path as osp
import time
from collections import OrderedDict, namedtuple

import numpy as np
import torch
from torch.nn import functional as F

from . import _lib, dist as ddp
from .checkpoint import save_checkpoint
from .losses import CrossEntropyLoss, DeepSupervision, multiModalMarginLossNew
from .meters import AverageMeter, DeferredSummary, MetricMeter
from .metrics import accuracy, compute_distance_matrix, evaluate_rank
from .optim import FusedAdam, FusedSGD

_Entry = namedtuple("_Entry", "model optim sched")
_SUMMARY_3M = ('loss', 'LossX', 'LossM', 'accR', 'lossR', 'accN', 'lossN', 'accT', 'lossT')


def _hms(seconds):
    return str(datetime.timedelta(seconds=int(seconds)))


class Engine(object):
    """Base driver.  Subclasses set self.model / optimizer / scheduler, register them and implement
    forward_backward(data) -> dict of scalars."""

    def __init__(self, datamanager, use_gpu=True):
        self.datamanager = datamanager
        self.train_loader = datamanager.train_loader
        self.test_loader = datamanager.test_loader
        self.use_gpu = bool(use_gpu) and torch.cuda.is_available()
        self.writer = None
        self.epoch = 0
        self.model = self.optimizer = self.scheduler = None
        self._registry = OrderedDict()       # name -> _Entry
        self._replicas_synced = False
        # data parallel: True when every rank's loader already yields that rank's shard (bench.py; a loader built by
        # ieee_amd.data.build_loaders says so itself through the batch's `global_rows`); otherwise each rank is handed
        # the global batch and keeps its identity-aligned slice (dist.shard_batch).  dp_total_rows: the global batch
        # size of a presharded loader that does not stamp its batches (None: one tiny all-reduce per step asks).
        self.dp_presharded = False
        self.dp_total_rows = None
        # True only when the SAME batch object is stepped over and over (bench.py): lets the 3M chunk check run once
        self.resident_batch = False

    # ---- registry ------------------------------------------------------------------------------------------------
    def register_model(self, name='model', model=None, optim=None, sched=None):
        self._registry[name] = _Entry(model, optim, sched)

    # the three dicts the reference keeps (engine.py:48-50), as read-only views of the registry
    @property
    def _models(self):
        return OrderedDict((k, e.model) for k, e in self._registry.items())

    @property
    def _optims(self):
        return OrderedDict((k, e.optim) for k, e in self._registry.items())

    @property
    def _scheds(self):
        return OrderedDict((k, e.sched) for k, e in self._registry.items())

    def get_model_names(self, names=None):
        if names is None:
            return list(self._registry)
        wanted = names if isinstance(names, list) else [names]
        for n in wanted:
            assert n in self._registry
        return wanted

    def set_model_mode(self, mode='train', names=None):
        assert mode in ['train', 'eval', 'test']
        for n in self.get_model_names(names):
            self._registry[n].model.train(mode == 'train')

    def get_current_lr(self, names=None):
        first = self.get_model_names(names)[0]
        return self._registry[first].optim.param_groups[-1]['lr']

    def update_lr(self, names=None):
        for n in self.get_model_names(names):
            sched = self._registry[n].sched
            if sched is not None:
                sched.step()

    def save_model(self, epoch, mAP, save_dir, is_best=False):
        """one file per registered model: <save_dir>/<name>/model.pth.tar-<epoch+1> (engine.py:77-101); rank 0 writes"""
        if ddp.rank() != 0:
            return
        for n, e in self._registry.items():
            state = {'state_dict': e.model.state_dict(), 'epoch': epoch + 1, 'mAP': mAP,
                     'optimizer': e.optim.state_dict(),
                     'scheduler': None if e.sched is None else e.sched.state_dict()}
            save_checkpoint(state, osp.join(save_dir, n), is_best=is_best)

    # ---- the loop ------------------------------------------------------------------------------------------------
    def run(self, save_dir='log', max_epoch=0, start_epoch=0, print_freq=10, fixbase_epoch=0, open_layers=None,
            start_eval=0, eval_freq=-1, test_only=False, dist_metric='euclidean', normalize_feature=False,
            visrank=False, visrank_topk=10, use_metric_cuhk03=False, ranks=[1, 5, 10, 20], rerank=False):
        """engine.py:126-232.  As in the reference there is NO evaluation or checkpoint after the last epoch (the
        `(epoch + 1) != max_epoch` guard, :216), and re-ranking applies to test_only runs (its docstring, :171-172;
        the in-loop test call does not pass it on, :217-225)."""
        if visrank and not test_only:
            raise ValueError('visrank can be set to True only if test_only=True')
        if visrank:
            raise NotImplementedError('visrank (the cv2 / t-SNE visualisation) is outside the hot path')
        eval_args = dict(dist_metric=dist_metric, normalize_feature=normalize_feature, save_dir=save_dir,
                         use_metric_cuhk03=use_metric_cuhk03, ranks=ranks)
        if test_only:
            self.test(rerank=rerank, **eval_args)
            return
        began = time.time()
        self.start_epoch, self.max_epoch = start_epoch, max_epoch
        print('=> Start training')
        for self.epoch in range(start_epoch, max_epoch):
            lap = time.time()
            self.train(print_freq=print_freq, fixbase_epoch=fixbase_epoch, open_layers=open_layers)
            print("Epoch Time: {}\t Total Time: {}\n".format(_hms(time.time() - lap), _hms(time.time() - began)))
            done = self.epoch + 1
            due = eval_freq > 0 and done >= start_eval and done % eval_freq == 0
            if due and done != max_epoch:
                mAP = self.test(**eval_args)
                self.save_model(self.epoch, mAP, save_dir)
        print('Elapsed {}'.format(str(datetime.timedelta(seconds=round(time.time() - began)))))
        if self.writer is not None:
            self.writer.close()

    def _warn_forking_loader(self):
        """The reference's data manager builds `DataLoader(num_workers=workers)` with the platform's default start method
        (torchreid/data/datamanager.py:214-229; scripts/default_config.py:20: workers = 1 by default) -- on Linux that fork()s
        the training process at the start of every epoch.  Beside a live HIP context the fork write-protects the page tables,
        the driver re-validates the process's GPU-visible host memory, and the queues stand still meanwhile: measured
        364 ms per train step (instead of 14.5) while a forked child lives, 78 ms per 1 KB copy (LABNOTES.md, "fork() beside a
        live HIP context").  Warn once, with the fix."""
        if getattr(self, "_fork_warned", False):
            return
        loader = self.train_loader
        workers = getattr(loader, "num_workers", 0)
        if not isinstance(loader, torch.utils.data.DataLoader) or not workers:
            return
        ctx = getattr(loader, "multiprocessing_context", None)
        if ctx is not None:
            method = ctx.get_start_method() if hasattr(ctx, "get_start_method") else str(ctx)
        else:
            import multiprocessing
            method = multiprocessing.get_start_method(allow_none=True) or ("fork" if os.name == "posix" and os.uname().sysname == "Linux" else "spawn")
        if method != "fork" or not (torch.cuda.is_available() and torch.cuda.is_initialized()):
            return
        self._fork_warned = True
        import warnings
        warnings.warn(
            "ieee_amd: the train loader is a torch DataLoader with %d worker process(es) started by fork() while this process "
            "holds a live HIP context. Every fork stalls the GPU queues of the training process (measured on MI355X: 364 ms per "
            "train step instead of 14.5 while a forked worker lives). Build the loader with "
            "multiprocessing_context='forkserver' (or 'spawn') and persistent_workers=True, or use "
            "ieee_amd.data.build_loaders(...), whose workers come from a fork server and only decode." % workers,
            RuntimeWarning, stacklevel=3)

    def train(self, print_freq=10, fixbase_epoch=0, open_layers=None):
        """one epoch (engine.py:234-282): forward_backward per batch, a report every print_freq batches, then the
        scheduler step"""
        log, t_batch, t_data = MetricMeter(), AverageMeter(), AverageMeter()
        self._warn_forking_loader()
        self.set_model_mode('train')
        self.two_stepped_transfer_learning(self.epoch, fixbase_epoch, open_layers)
        self.num_batches = len(self.train_loader)
        mark = time.time()
        # the per-batch numbers are read back when the log is printed (or written), not inside every step
        self.defer_summary = os.environ.get("IEEE_DEFER_SUMMARY", "1") != "0"
        for self.batch_idx, data in enumerate(self.train_loader):
            t_data.update(time.time() - mark)
            log.update(self.forward_backward(data))
            t_batch.update(time.time() - mark)
            seen = self.batch_idx + 1
            if seen % print_freq == 0 and ddp.rank() == 0:
                print('epoch: [{}/{}][{}/{}]\tlr {:.6f}\n{}\t'.format(self.epoch + 1, self.max_epoch, seen, self.num_batches,
                                                                      self.get_current_lr(), log))
            if self.writer is not None:
                step = self.epoch * self.num_batches + self.batch_idx
                self.writer.add_scalar('Train/time', t_batch.avg, step)
                self.writer.add_scalar('Train/data', t_data.avg, step)
                for key, meter in log.meters.items():
                    self.writer.add_scalar('Train/' + key, meter.avg, step)
                self.writer.add_scalar('Train/lr', self.get_current_lr(), step)
            mark = time.time()
        self.defer_summary = False
        str(log)                                 # settle what is still pending before the epoch ends
        self.update_lr()

    def forward_backward(self, data):
        raise NotImplementedError

    # ---- evaluation ----------------------------------------------------------------------------------------------
    def test(self, dist_metric='euclidean', normalize_feature=False, visrank=False, visrank_topk=10, save_dir='',
             use_metric_cuhk03=False, ranks=[1, 5, 10, 20], rerank=False):
        """every target dataset in turn (engine.py:287-337); returns the last one's mAP like the reference"""
        self.set_model_mode('eval')
        mAP = 0.0
        for name, loaders in self.test_loader.items():
            where = 'source' if name in self.datamanager.sources else 'target'
            print('##### Evaluating {} ({}) #####'.format(name, where))
            rank1, mAP = self._evaluate(dataset_name=name, query_loader=loaders['query'],
                                        gallery_loader=loaders['gallery'], dist_metric=dist_metric,
                                        normalize_feature=normalize_feature, use_metric_cuhk03=use_metric_cuhk03,
                                        ranks=ranks, rerank=rerank)
            if self.writer is not None:
                self.writer.add_scalar('Test/{}/rank1'.format(name), rank1, self.epoch)
                self.writer.add_scalar('Test/{}/mAP'.format(name), mAP, self.epoch)
        return mAP

    def _descriptors(self, loader, clock):
        """[rows, 2304] descriptors of one loader, on the device, with the identity / camera labels.  With several
        ranks each one runs the forward for every world-th batch and one all-gather completes the matrix."""
        world, me = ddp.world_size(), ddp.rank()
        mine, rows, pids, cams = {}, [], [], []
        # a loader that is itself sharded over the ranks (ieee_amd.data.DeviceLoader(rank=, world=)) decodes only this
        # rank's batches and knows every batch's labels from its records; any other loader is walked in full and the
        # foreign batches are skipped before the forward
        presharded = world > 1 and getattr(loader, "sharded", False) and getattr(loader, "world", 1) == world
        if presharded:
            for p, c in loader.batch_labels():
                pids.append(np.asarray(p).reshape(-1))
                cams.append(np.asarray(c).reshape(-1))
                rows.append(len(p))
        for b, data in enumerate(loader):
            imgs, p, c, timeids = self.parse_data_for_eval(data)
            if presharded:
                b = int(data['batch_index'])
            else:
                p, c = np.asarray(p).reshape(-1), np.asarray(c).reshape(-1)
                pids.append(p)
                cams.append(c)
                rows.append(len(p))
                if b % world != me:
                    continue
            if self.use_gpu:
                imgs = [im.cuda(non_blocking=True) for im in imgs]
            t0 = time.time()
            mine[b] = self.extract_features(imgs, timeids).clone()
            clock.update(time.time() - t0)
        if world == 1:
            feats = torch.cat([mine[b] for b in range(len(rows))], 0)
        else:
            some = next(iter(mine.values())) if mine else None
            width = some.shape[1] if some is not None else 2304
            device = some.device if some is not None else torch.device('cuda' if self.use_gpu else 'cpu')
            feats = ddp.gather_feature_batches(mine, rows, width, device)
        return feats, np.concatenate(pids), np.concatenate(cams)

    @torch.no_grad()
    def _evaluate(self, dataset_name='', query_loader=None, gallery_loader=None, dist_metric='euclidean',
                  normalize_feature=False, visrank=False, visrank_topk=10, save_dir='', use_metric_cuhk03=False,
                  ranks=[1, 5, 10, 20], rerank=False):
        """descriptors -> distance matrix -> CMC / mAP, printed like engine.py:339-441; returns (rank-1, mAP)"""
        clock = AverageMeter()
        say = print if ddp.rank() == 0 else (lambda *a, **k: None)     # one report, not one per rank
        if ddp.world_size() > 1 and self.model is not None and hasattr(self.model, "_flat_buffers"):
            ddp.sync_replicas(self.model, buffers_only=True)       # DataParallel evaluates with GPU 0's statistics
        say('Extracting features from query set ...')
        qf, q_pids, q_camids = self._descriptors(query_loader, clock)
        say('Done, obtained {}-by-{} matrix'.format(qf.size(0), qf.size(1)))
        say('Extracting features from gallery set ...')
        gf, g_pids, g_camids = self._descriptors(gallery_loader, clock)
        say('Done, obtained {}-by-{} matrix'.format(gf.size(0), gf.size(1)))
        say('Speed: {:.4f} sec/batch'.format(clock.avg))
        if normalize_feature:
            say('Normalzing features with L2 norm ...')
            qf, gf = F.normalize(qf, p=2, dim=1), F.normalize(gf, p=2, dim=1)
        say('Computing distance matrix with metric={} ...'.format(dist_metric))
        sharded = ddp.world_size() > 1 and not use_metric_cuhk03 and not rerank
        if sharded:     # each rank ranks its slice of the queries against the whole gallery (ieee_amd/dist.py)
            say('Computing CMC and mAP for {} (queries sharded over {} ranks)'.format(dataset_name, ddp.world_size()))
            cmc, mAP = ddp.sharded_evaluate_rank(qf, gf, q_pids, g_pids, q_camids, g_camids, metric=dist_metric)
        else:
            distmat = compute_distance_matrix(qf, gf, dist_metric)
            if rerank:  # engine.py:402-406: k-reciprocal re-ranking with the query-query and gallery-gallery matrices
                say('Applying person re-ranking ...')
                from .rerank import re_ranking
                distmat = re_ranking(distmat, compute_distance_matrix(qf, qf, dist_metric),
                                     compute_distance_matrix(gf, gf, dist_metric))
            say('Computing CMC and mAP for {}'.format(dataset_name))
            cmc, mAP = evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, use_metric_cuhk03=use_metric_cuhk03)
        say('** Results **')
        say('mAP: {:.2%}'.format(mAP))
        say('CMC curve')
        for r in ranks:
            say('Rank-{:<3}: {:.2%}'.format(r, cmc[r - 1]))     # IndexError when r exceeds the curve, as in the reference
        say('\n')
        return cmc[0], mAP

    # ---- small hooks the reference exposes ------------------------------------------------------------------------
    def compute_loss(self, criterion, outputs, targets):
        if isinstance(outputs, (tuple, list)):
            return DeepSupervision(criterion, outputs, targets)
        return criterion(outputs, targets)

    def extract_features(self, input, timeids):
        return self.model(input, timeids)

    def parse_data_for_train(self, data):
        return data['img'], data['pid'], data['timeid']

    def parse_data_for_eval(self, data):
        return data['img'], data['pid'], data['camid'], data['timeid']

    def two_stepped_transfer_learning(self, epoch, fixbase_epoch, open_layers, model=None):
        """Freeze everything but `open_layers` for the first fixbase_epoch epochs, then train all (engine.py:507-529,
        utils/torchtools.py:160-221: open_specified_layers puts every other child in eval() mode and stops its gradients,
        open_all_layers undoes both).  The shipped configuration has fixbase_epoch = 0.  A native model takes the frozen
        children through set_frozen_children: their BatchNorms then run on the running statistics inside the native
        training forward and the backward differentiates through them as fixed affine maps."""
        model = self.model if model is None else model
        if model is None:
            return
        freezing = (epoch + 1) <= fixbase_epoch and open_layers is not None
        keep = []
        if freezing:
            keep = [open_layers] if isinstance(open_layers, str) else list(open_layers)
            for layer in keep:
                assert hasattr(model, layer), \
                    '"{}" is not an attribute of the model, please provide the correct name'.format(layer)
            print('* Only train {} (epoch: {}/{})'.format(keep, epoch + 1, fixbase_epoch))
        frozen = []
        for child_name, child in model.named_children():
            flag = (child_name in keep) if freezing else True
            if not flag:
                frozen.append(child_name)
            if not _is_native(model):
                child.train(flag)                     # a plain nn.Module: eval() mode for the frozen children
            for p in child.parameters():
                p.requires_grad = flag
        if _is_native(model):
            model.set_frozen_children(frozen)

    # ---- data parallel helpers ---------------------------------------------------------------------------------------
    def _local_batch(self, data):
        """(this rank's slice of the batch, rows of the GLOBAL batch)"""
        world = ddp.world_size()
        rows = int(data['pid'].shape[0])
        if world == 1:
            return data, rows
        stamped = data.get('global_rows') if isinstance(data, dict) else None
        if stamped is not None:                    # a shard-aware loader: this rank's rows of a global batch of `stamped`
            return data, int(stamped)
        if self.dp_presharded:
            return data, int(self.dp_total_rows) if self.dp_total_rows is not None else ddp.global_rows(rows)
        k = int(getattr(self.datamanager, 'num_instances', 4))
        return ddp.shard_batch(data, k), rows

    def _sync_replicas_once(self):
        """before the first step every rank takes rank 0's parameters, buffers and optimizer state: native models through
        their flat buffers, any other nn.Module / torch.optim optimizer tensor by tensor (dist.sync_replicas)"""
        if ddp.world_size() > 1 and not self._replicas_synced and self.model is not None:
            ddp.sync_replicas(self.model, self.optimizer)
        self._replicas_synced = True


def _is_native(model):
    return hasattr(model, "native_net")


def _chunks_short_of_identities(pids):
    """True when `feat.chunk(n)` yields fewer than n = len(unique(pids)) pieces, i.e. when the reference's 3M loss dies
    with `IndexError: tuple index out of range` in its first loop (multi_modal_margin_loss_new.py:24-33) BEFORE any
    gradient exists.  Evaluated on the host so that the same error can be raised before this step touches the weights."""
    rows = int(pids.numel())
    n = int(torch.unique(pids).numel())
    per = -(-rows // n)
    return -(-rows // per) < n


class _FusedStepMixin(object):
    """the native train step shared by the 3M and the CE-only engines"""

    def _fused_ok(self):
        return _is_native(self.model) and isinstance(self.optimizer, (FusedSGD, FusedAdam)) and self.use_gpu

    def _guard_chunks(self, pids, weight_m):
        """the reference's IndexError, raised where the reference raises it: before backward / the optimizer"""
        if weight_m <= 0:
            return
        # checked on every batch (64 labels: negligible next to a step); only a caller that declares its batch resident
        # (`engine.resident_batch`, bench.py: the same tensor object every step) gets the answer of the first check --
        # the address / version / shape of a DataLoader batch is NOT an identity: the allocator hands the same block out
        # again for the next batch
        if self.resident_batch and getattr(self, "_chunk_for", None) is pids:
            bad = self._chunk_bad
        else:
            bad = _chunks_short_of_identities(pids)
            self._chunk_for, self._chunk_bad = (pids if self.resident_batch else None), bad
        if bad:
            raise IndexError('tuple index out of range')

    def _fused_step(self, imgs, pids, weight_x, weight_m, margin, eps, total_rows=None):
        lib = _lib.require_gpu()
        m = self.model
        B, _, H, W = imgs[0].shape
        dev = m._flat_params.device
        net = self._step_net = m.native_net(B, H, W)
        pids = pids.to(device=dev, dtype=torch.int64).contiguous()
        m._bump_counters()
        # Range guard of the bf16 BatchNorm totals, acted on ON THE DEVICE (single GPU, FusedSGD): a step whose sums were clamped
        # is skipped -- the optimizer launches take the executor's flag words as `skip_flags`, and the running statistics the
        # forward wrote are put back from a copy (ieee_guard_buffers) -- so that the engine, when it reads the report with the
        # step's summary, can switch to the partial-sum path and go on from an undamaged state (_check_bn_range).
        staged = ddp.world_size() > 1 or (os.environ.get("IEEE_FORCE_DP_PATH") == "1" and torch.distributed.is_initialized())
        guard = (not staged) and isinstance(self.optimizer, FusedSGD) and net.dtype == torch.bfloat16
        self._step_guarded = guard
        flags = net.flags_view() if net.dtype == torch.bfloat16 else None
        self._step_flags = flags
        if isinstance(self.optimizer, FusedSGD):
            self.optimizer.skip_flags = flags if guard else None
        if guard:
            bk = getattr(self, "_buf_backup", None)
            if bk is None or bk.shape != m._flat_buffers.shape or bk.device != m._flat_buffers.device or \
                    self._buf_backup_key != (m._flat_buffers._version, id(m)):
                bk = self._buf_backup = m._flat_buffers.detach().clone()      # (again after a load_state_dict / a replica sync)
                self._buf_backup_key = (m._flat_buffers._version, id(m))
        logits, feats = net.forward(imgs, training=True)
        C = logits.shape[2]
        if not hasattr(self, "_scratch") or self._scratch[0].shape != logits.shape:
            self._scratch = (torch.empty_like(logits), torch.empty_like(feats),
                             torch.empty(18 * 2 + 3, dtype=torch.float32, device=dev),
                             torch.empty(18 * B * 2, dtype=torch.float32, device=dev),
                             torch.empty(B + 3, dtype=torch.float32, device=dev))
        dl, df, small, work, mwork = self._scratch
        head_loss, head_acc, out3 = small[:18], small[18:36], small[36:39]
        ce_scale = float(weight_x) * ddp.ce_grad_scale(B, total_rows)
        _lib.check(lib.ieee_ce_ls_fwd_bwd(_lib.ptr(logits), _lib.ptr(pids), _lib.ptr(dl), _lib.ptr(head_loss),
                                          _lib.ptr(head_acc), _lib.ptr(work), 18, B, C, float(eps), ce_scale,
                                          _lib.stream()))
        if weight_m > 0:
            _lib.check(lib.ieee_margin3m_fwd_bwd(_lib.ptr(feats), _lib.ptr(pids), _lib.ptr(df), _lib.ptr(out3),
                                                 _lib.ptr(mwork), B, feats.shape[2], float(margin), float(weight_m),
                                                 _lib.stream()))
        else:
            df.zero_()
            out3.zero_()
        def guard_buffers():     # running statistics: back to the copy when the forward clamped, else the copy follows them
            if guard:
                _lib.check(lib.ieee_guard_buffers(_lib.ptr(flags), _lib.ptr(m._flat_buffers), _lib.ptr(self._buf_backup),
                                                  m._flat_buffers.numel(), _lib.stream()))
        if not staged and isinstance(self.optimizer, FusedSGD) and os.environ.get("IEEE_OPT_OVERLAP", "1") != "0":
            # single GPU: parts 0-3 (head, layer4, layer3, layer2 = 90 % of the parameters) are updated on a helper
            # stream while the compute stream still runs the backward of layer1 + stem and the side stream drains the
            # last weight gradients; only layer1 + stem wait for the final join.  Same arithmetic as optimizer.step().
            main = torch.cuda.current_stream()
            if not hasattr(self, "_opt_stream"):
                self._opt_stream = torch.cuda.Stream(priority=int(os.environ.get("IEEE_OPT_PRIO", "0")))
            helper = self._opt_stream
            for part in range(4):
                net.backward_part_async(dl, df, part)
            helper.wait_stream(main)
            net.side_wait(helper)
            net.backward_part_async(dl, df, 4)
            with torch.cuda.stream(helper):
                guard_buffers()                  # (the forward's flags are final; off the critical path)
                for part in range(4):
                    self.optimizer.step_part(part)
            net.side_wait()
            self.optimizer.step_part(4)
            main.wait_stream(helper)
            return small, out3
        # IEEE_DP_OVERLAP=0 / engine.dp_overlap = False: the plain data-parallel step of the reference's DataParallel run
        # (scripts/mainMultiModal.py:219-220) -- whole backward, ONE all-reduce pass over the flat gradient on the compute
        # stream, one optimizer step.  No helper stream, nothing overlapped: the fallback when the overlapped form below
        # meets a stream -> hardware-queue assignment it does not like (bench.py times both at N > 1 and says which it used).
        overlap = getattr(self, "dp_overlap", None)
        if overlap is None:
            overlap = os.environ.get("IEEE_DP_OVERLAP", "1") != "0"
        if staged and not overlap:
            net.backward(dl, df)
            self._allreduce_ranges(m, [r for ranges in m.grad_part_ranges() for r in ranges])
        elif not staged:
            net.backward(dl, df)
        else:
            # data parallel: the backward runs in 5 parts; as soon as a part is done its (final) slice of the flat
            # gradient is all-reduced (sum) asynchronously over RCCL/xGMI while the next part computes.  Together
            # the slices are exactly one pass over the 438 MB buffer; torch orders each collective after the
            # kernels already enqueued on the compute stream, and wait() orders the SGD step after them.
            # The compute stream is never blocked between parts: a helper stream waits for the part's kernels on the
            # compute stream AND for its weight gradients on the executor's side stream, and the collective is issued
            # from there.
            # The optimizer update of a part (same arithmetic as optimizer.step(), FusedSGD) follows its collective on that
            # helper stream too, so only the last part's update is left when the backward ends (it was one 0.7 ms update of
            # all parameters on the compute stream after the last collective).
            main = torch.cuda.current_stream()
            if not hasattr(self, "_comm_stream"):
                # high priority in a real data-parallel job (with ONE hardware queue per priority, ieee_amd/__init__.py, it
                # then shares torch's collective stream's queue and never the compute stream's: LABNOTES.md section 6); normal
                # priority in the single-GPU 1-rank leg of bench.py, where two queues per priority are on
                prio = os.environ.get("IEEE_COMM_PRIO")
                # (with two or more queues per priority a busy high-priority communication stream beside torch's busy
                # collective stream is the one combination that fell apart in the probe: high priority only with ONE queue)
                from . import HW_QUEUES         # what the runtime really runs with (not what the environment says now)
                prio = int(prio) if prio is not None else (-1 if ddp.world_size() > 1 and HW_QUEUES == 1 else 0)
                self._comm_stream = torch.cuda.Stream(priority=prio)
            comm = self._comm_stream
            by_part = isinstance(self.optimizer, FusedSGD) and os.environ.get("IEEE_OPT_OVERLAP", "1") != "0"
            for part, ranges in enumerate(m.grad_part_ranges()):
                net.backward_part_async(dl, df, part)
                comm.wait_stream(main)
                net.side_wait(comm)
                with torch.cuda.stream(comm):
                    timing = getattr(self, "time_collectives", None)     # bench.py's RCCL leg: event pair per part
                    if timing is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(comm)
                    self._allreduce_ranges(m, ranges)
                    if timing is not None:
                        e1.record(comm)
                        timing.append((part, e0, e1, (4 if self._grad_dtype() == 'fp32' else 2) * sum(b - a for a, b in ranges)))
                    if by_part:
                        self.optimizer.step_part(part)
            net.side_wait()                      # final join of the weight-gradient stream into the compute stream
            main.wait_stream(comm)
            if by_part:
                return small, out3
        guard_buffers()
        self.optimizer.step()
        return small, out3

    def _grad_dtype(self):
        """'fp32' (default: what the reference's DataParallel reduces) or 'bf16' (`engine.dp_grad_dtype` /
        IEEE_DP_GRAD_DTYPE=bf16: half the bytes on the wire; every rank's slice is rounded to bf16 before the sum and the sum
        itself is a bf16 sum -- relative error of a reduced element <= ~2^-8 * (1 + log2 world), tests/test_dist_cpu.py)"""
        v = getattr(self, "dp_grad_dtype", None) or os.environ.get("IEEE_DP_GRAD_DTYPE", "fp32")
        v = {"float32": "fp32", "bfloat16": "bf16"}.get(v, v)
        if v not in ("fp32", "bf16"):
            raise ValueError("IEEE_DP_GRAD_DTYPE / dp_grad_dtype must be fp32 or bf16, not %r" % (v,))
        return v

    def _allreduce_ranges(self, m, ranges):
        """all-reduce (sum) of these [a, b) slices of the flat gradient, enqueued on the CURRENT stream; on return that
        stream (not the host) is ordered after the collectives"""
        ar = torch.distributed.all_reduce
        if self._grad_dtype() == "fp32":
            handles = [ar(m._flat_grads[a:b], op=torch.distributed.ReduceOp.SUM, async_op=True) for a, b in ranges]
            for h in handles:
                h.wait()
            return
        lib = _lib.load()
        stage = getattr(m, "_flat_grads_bf16", None)
        if stage is None or stage.numel() != m._flat_grads.numel() or stage.device != m._flat_grads.device:
            stage = m._flat_grads_bf16 = torch.empty(m._flat_grads.numel(), dtype=torch.bfloat16, device=m._flat_grads.device)
        for a, b in ranges:
            _lib.check(lib.ieee_grad_pack_bf16(_lib.ptr(m._flat_grads[a:b]), _lib.ptr(stage[a:b]), b - a, _lib.stream()))
        handles = [ar(stage[a:b], op=torch.distributed.ReduceOp.SUM, async_op=True) for a, b in ranges]
        for h in handles:
            h.wait()
        for a, b in ranges:
            _lib.check(lib.ieee_grad_unpack_bf16(_lib.ptr(stage[a:b]), _lib.ptr(m._flat_grads[a:b]), b - a, _lib.stream()))

    _RING = 4                            # pinned read-back buffers: the host runs at most _RING - 1 steps ahead

    def _summary_from(self, small, finish, keys):
        """The step's single device->host read-back (39 floats into a pinned buffer, enqueued on the step's stream behind
        its last kernel; a pageable .cpu() adds ~50 us of staging) and `finish(numbers)` -> the loss_summary values.
        Eager (default, the reference's behaviour): wait for it and return a plain dict.  `self.defer_summary`: return a
        DeferredSummary that waits when first looked at, so the host can enqueue the next step while this one runs (the
        end-of-step bubble -- synchronise, read, re-launch -- was ~0.2 ms of a 16.5 ms step)."""
        ring = getattr(self, "_small_ring", None)
        if ring is None or ring[0][0].numel() != small.numel():
            ring = self._small_ring = [[torch.empty(small.numel(), dtype=small.dtype, pin_memory=True), None]
                                       for _ in range(self._RING)]
            self._ring_at = 0
        slot = ring[self._ring_at]
        self._ring_at = (self._ring_at + 1) % self._RING
        if slot[1] is not None:          # a summary nobody has looked at still owns this buffer: settle it first
            slot[1].resolve()
        host = slot[0]
        host.copy_(small, non_blocking=True)
        flags, hostf = getattr(self, "_step_flags", None), None
        if flags is not None:                # the step's range-guard words ride along (16 bytes, same stream, same event)
            if len(slot) < 3:
                slot.append(torch.zeros(4, dtype=torch.int32, pin_memory=True))
            hostf = slot[2]
            hostf.copy_(flags, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(small.device))

        net = getattr(self, "_step_net", None)
        step_idx = self._steps_done = getattr(self, "_steps_done", 0) + 1      # 1-based index of this engine's train steps
        guarded = getattr(self, "_step_guarded", False)

        def read():
            done.synchronize()
            slot[1] = None
            if hostf is not None:
                self._check_bn_range(net, step_idx, tuple(int(v) for v in hostf.tolist()), guarded)
            v = host.numpy().copy()
            hl, ha = v[:18], v[18:36]
            lR, lN, lT = float(hl[0:6].sum()), float(hl[6:12].sum()), float(hl[12:18].sum())
            aR, aN, aT = float(ha[0:6].mean()), float(ha[6:12].mean()), float(ha[12:18].mean())
            return dict(zip(keys, finish(lR, lN, lT, aR, aN, aT, float(v[36]))))

        if not getattr(self, "defer_summary", False):
            return read()
        slot[1] = DeferredSummary(keys, read)
        return slot[1]

    def _check_bn_range(self, net, step, flags, guarded):
        """The bf16 train step keeps its BatchNorm sums as int64 fixed-point totals (include/ieee_amd.h,
        ieee_conv_next_bn_totals): bit-reproducible, but with a RANGE (forward sum y^2 up to 2.7e11 per channel, backward sums up
        to 4.2e6) that torch's fp32 batch_norm (reference: torchreid/models/resnet.py:164-184) does not have.  The kernels
        clamp and report instead of wrapping; `flags` are the step's four report words, copied with its summary.
        A clamped tile means the statistics of that step were not fp32 BatchNorm's.  `guarded` (single GPU, FusedSGD): the step
        has been SKIPPED on the device -- the optimizer launches saw the words and left parameters, momentum and shadow alone,
        the running statistics were put back -- so the engine DEGRADES and goes on: the executor is switched to the per-tile
        partial-sum path (fp32 sums, no range: the reference's semantics at any magnitude, ieee_net_set_bn_totals) for every
        following step, with one RuntimeWarning that names the step.  Not guarded (data parallel: a rank-local skip would split
        the replicas; another optimizer), or IEEE_BN_STRICT=1: raise, as round 5 did.  A total beyond half the range is still
        exact: warn once."""
        if net is None:
            return
        f_clamp, b_clamp, f_half, b_half = flags
        if f_clamp or b_clamp:
            which = " and ".join(w for w, on in (("forward (sum y, sum y^2 of a conv output)", f_clamp),
                                                 ("backward (sum g, sum g*y)", b_clamp)) if on)
            msg = ("BatchNorm statistics left the range of the fixed-point totals in the %s pass of step %s (or were NaN): "
                   "the int64 totals hold sum y^2 up to 2.7e11 and backward sums up to 4.2e6 per channel; beyond that a tile sum is "
                   "clamped, so the statistics -- and everything computed from them -- are not what fp32 BatchNorm gives. "
                   "Activations / gradients of that size usually mean the run is diverging." % (which, "?" if step is None else step))
            import os
            if os.environ.get("IEEE_BN_STRICT", "0") == "1" or not guarded:
                raise _lib.IeeeAmdError(msg + ("  (IEEE_BN_STRICT=1: raising.)" if guarded else "  The update of that step HAS BEEN APPLIED (no "
                                               "device-side skip in this configuration): restart from the last checkpoint.") +
                                        "  IEEE_BN_TOTALS_TILES=0 selects the per-tile partial-sum path, which has no range limit, "
                                        "from the start.")
            net.set_bn_totals(False)
            self.model._bn_totals_off = True               # executors built later (another batch shape) start degraded too
            if not getattr(self, "_bn_degraded", False):
                import warnings
                self._bn_degraded = True
                warnings.warn("ieee_amd: %s  That step was SKIPPED (parameters, momentum and running statistics are those before it), "
                              "and so are the steps already queued behind it that clamp as well (at most %d).  Switched to the "
                              "per-tile partial-sum path (fp32 sums, no range limit; about 0.2 ms per step slower) for every "
                              "following step; IEEE_BN_STRICT=1 raises here instead." % (msg, self._RING - 1), RuntimeWarning)
            return
        if (f_half or b_half) and not getattr(self, "_bn_range_warned", False):
            import warnings
            self._bn_range_warned = True
            warnings.warn("ieee_amd: a BatchNorm total of the bf16 train step is beyond half the range of its int64 fixed point "
                          "(%s); the statistics are still exact, but a further doubling of the %s would be clamped (that step would "
                          "be skipped and the engine would fall back to the partial-sum path; IEEE_BN_TOTALS_TILES=0 selects it from "
                          "the start)" % ("forward" if f_half else "backward", "activations" if f_half else "gradients"))

    def _generic_allreduce(self, params):
        """autograd path under data parallelism: sum the parameter gradients across ranks (one collective per tensor;
        the fused path is the fast one)"""
        if ddp.world_size() == 1:
            return
        for p in params:
            if p.grad is not None:
                torch.distributed.all_reduce(p.grad, op=torch.distributed.ReduceOp.SUM)

    def _to_device(self, imgs, pids, pids_dev=None):
        """pids_dev: the labels already on the device (ieee_amd.data.DeviceLoader stamps `pid_dev` on its batches): a
        `pids.cuda()` from pageable memory blocks the host until the launch stream has drained"""
        if self.use_gpu:
            imgs = [im.cuda(non_blocking=True) for im in imgs]
            pids = pids_dev if pids_dev is not None else pids.cuda()
        return imgs, pids


class Image3MEngine(_FusedStepMixin, Engine):
    """Cross entropy over the 18 heads (label smoothed) + the 3M margin loss: engine/image/margin.py:62-154."""

    def __init__(self, datamanager, model, optimizer, margin=3, weight_m=1, weight_x=1, scheduler=None, use_gpu=True,
                 label_smooth=True):
        super(Image3MEngine, self).__init__(datamanager, use_gpu)
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.register_model('model', model, optimizer, scheduler)
        assert weight_m >= 0 and weight_x >= 0
        assert weight_m + weight_x > 0
        self.weight_m, self.weight_x, self.margin = weight_m, weight_x, margin
        self.criterion_m = multiModalMarginLossNew(margin=margin)
        self.criterion_x = CrossEntropyLoss(num_classes=self.datamanager.num_train_pids, use_gpu=self.use_gpu,
                                            label_smooth=label_smooth)

    def forward_backward(self, data):
        self._sync_replicas_once()
        data, total_rows = self._local_batch(data)
        imgs, pids, timeids = self.parse_data_for_train(data)
        self._guard_chunks(pids, self.weight_m)
        imgs, pids = self._to_device(imgs, pids, data.get('pid_dev') if isinstance(data, dict) else None)
        if self._fused_ok():
            small, out3 = self._fused_step(imgs, pids, self.weight_x, self.weight_m, self.margin, self.criterion_x.eps,
                                           total_rows)
            loss_m = out3[0].clone()     # 'LossM' stays a 0-d device tensor like the reference's (margin.py:145)

            def finish(lR, lN, lT, aR, aN, aT, lm):
                loss_x = lR + lN + lT
                return (self.weight_m * lm + self.weight_x * loss_x, loss_x, loss_m, aR, lR, aN, lN, aT, lT)

            return self._summary_from(small, finish, _SUMMARY_3M)
        # ---- generic path: the reference's step (margin.py:102-152) over the autograd bridge
        oR, oN, oT, fR, fN, fT = self.model(imgs)
        scale = ddp.ce_grad_scale(int(pids.shape[0]), total_rows)
        loss, loss_m = 0, 0
        if self.weight_m > 0:
            loss_m = self.criterion_m(fR, fN, fT, pids)
            loss = loss + self.weight_m * loss_m
        if self.weight_x > 0:
            per_modality = [self.compute_loss(self.criterion_x, o, pids) for o in (oR, oN, oT)]
            loss_x = per_modality[0] + per_modality[1] + per_modality[2]
            loss = loss + (self.weight_x * scale) * loss_x
        self.optimizer.zero_grad()
        loss.backward()
        self._generic_allreduce(self.model.parameters())
        self.optimizer.step()
        acc = [sum(accuracy(h, pids)[0] for h in heads) / len(heads) for heads in (oR, oN, oT)]
        shown = self.weight_m * loss_m + self.weight_x * loss_x          # the unscaled loss, as the reference logs it
        values = (float(shown), loss_x.item(), loss_m, acc[0].item(), per_modality[0].item(), acc[1].item(),
                  per_modality[1].item(), acc[2].item(), per_modality[2].item())
        return dict(zip(_SUMMARY_3M, values))


class MultiModalImageSoftmaxEngine(_FusedStepMixin, Engine):
    """Cross entropy only (the "3M off" leg of the ablation): engine/image/softmax.py:11-132."""

    def __init__(self, datamanager, model, optimizer, scheduler=None, use_gpu=True, label_smooth=True):
        super(MultiModalImageSoftmaxEngine, self).__init__(datamanager, use_gpu)
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.register_model('model', model, optimizer, scheduler)
        self.criterion = CrossEntropyLoss(num_classes=self.datamanager.num_train_pids, use_gpu=self.use_gpu,
                                          label_smooth=label_smooth)

    def forward_backward(self, data):
        self._sync_replicas_once()
        data, total_rows = self._local_batch(data)
        imgs, pids, timeids = self.parse_data_for_train(data)
        imgs, pids = self._to_device(imgs, pids, data.get('pid_dev') if isinstance(data, dict) else None)
        if self._fused_ok():
            small, _ = self._fused_step(imgs, pids, 1.0, 0.0, 0.0, self.criterion.eps, total_rows)
            return self._summary_from(small, lambda lR, lN, lT, aR, aN, aT, lm: (lR + lN + lT, lR, aR, lN, aN, lT, aT),
                                      ('loss_all', 'loss_R', 'acc_R', 'loss_N', 'acc_N', 'loss_T', 'acc_T'))
        else:
            heads = self.model(imgs)[:3]
            per_modality = [self.compute_loss(self.criterion, o, pids) for o in heads]
            loss = per_modality[0] + per_modality[1] + per_modality[2]
            self.optimizer.zero_grad()
            (loss * ddp.ce_grad_scale(int(pids.shape[0]), total_rows)).backward()
            self._generic_allreduce(self.model.parameters())
            self.optimizer.step()
            lR, lN, lT = (v.item() for v in per_modality)
            aR, aN, aT = (sum(accuracy(h, pids)[0] for h in o).item() / len(o) for o in heads)
        return {'loss_all': lR + lN + lT, 'loss_R': lR, 'acc_R': aR, 'loss_N': lN, 'acc_N': aN, 'loss_T': lT,
                'acc_T': aT}
