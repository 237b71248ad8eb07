"""Mirror of torchreid.engine for the hot path: the generic loop (reference torchreid/engine/engine.py:
Engine.run :126-232, train :234-282, test :287-337, _evaluate :339-441) and the two live image engines
(Image3MEngine, engine/image/margin.py:62-154; MultiModalImageSoftmaxEngine, engine/image/softmax.py:
11-132).  Same constructor / method names, argument meaning and loss_summary keys.

Two ways through `forward_backward`:
  * fused (native model + ieee_amd.optim.FusedSGD): forward, 18-head CE, 3M, backward, the single RCCL
    gradient all-reduce and the SGD step are each one native call; no autograd graph, one small
    device->host read-back for the logging dict (the reference does >= 27 host syncs per step).
  * generic (any other optimizer): autograd over the same native forward/backward + HIP-backed criteria.
Evaluation keeps features on the device and feeds the distance-matrix and CMC/mAP kernels directly
(the reference copies every feature batch to the host and runs both on the CPU, engine.py:368-417)."""
from __future__ import absolute_import, division, print_function

import datetime
import os
import os.path as osp
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch
from torch.nn import functional as F

from . import _lib, dist as ddp
from .losses import CrossEntropyLoss, DeepSupervision, multiModalMarginLossNew
from .metrics import accuracy, compute_distance_matrix, evaluate_rank
from .optim import FusedAdam, FusedSGD


class AverageMeter(object):
    """reference utils/avgmeter.py:8-31"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class MetricMeter(object):
    """reference utils/avgmeter.py:34-73 (tensors are .item()'d, :63-65)"""

    def __init__(self, delimiter='\t'):
        self.meters = defaultdict(AverageMeter)
        self.delimiter = delimiter

    def update(self, input_dict):
        if input_dict is None:
            return
        if not isinstance(input_dict, dict):
            raise TypeError('Input to MetricMeter.update() must be a dictionary')
        for k, v in input_dict.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(v)

    def __str__(self):
        return self.delimiter.join('{} {:.4f} ({:.4f})'.format(n, m.val, m.avg) for n, m in self.meters.items())


def save_checkpoint(state, save_dir, is_best=False, remove_module_from_keys=False):
    """reference utils/torchtools.py:20-58: <save_dir>/model.pth.tar-<epoch>"""
    os.makedirs(save_dir, exist_ok=True)
    if remove_module_from_keys:
        state['state_dict'] = OrderedDict((k[7:] if k.startswith('module.') else k, v)
                                          for k, v in state['state_dict'].items())
    fpath = osp.join(save_dir, 'model.pth.tar-' + str(state['epoch']))
    torch.save(state, fpath)
    print('Checkpoint saved to "{}"'.format(fpath))
    if is_best:
        import shutil
        shutil.copy(fpath, osp.join(osp.dirname(fpath), 'model-best.pth.tar'))


class Engine(object):
    def __init__(self, datamanager, use_gpu=True):
        self.datamanager = datamanager
        self.train_loader = self.datamanager.train_loader
        self.test_loader = self.datamanager.test_loader
        self.use_gpu = (torch.cuda.is_available() and use_gpu)
        self.writer = None
        self.epoch = 0
        self.model = None
        self.optimizer = None
        self.scheduler = None
        self._models = OrderedDict()
        self._optims = OrderedDict()
        self._scheds = OrderedDict()

    def register_model(self, name='model', model=None, optim=None, sched=None):
        self._models[name] = model
        self._optims[name] = optim
        self._scheds[name] = sched

    def get_model_names(self, names=None):
        names_real = list(self._models.keys())
        if names is not None:
            if not isinstance(names, list):
                names = [names]
            for name in names:
                assert name in names_real
            return names
        return names_real

    def save_model(self, epoch, mAP, save_dir, is_best=False):
        if ddp.rank() != 0:
            return
        for name in self.get_model_names():
            save_checkpoint({
                'state_dict': self._models[name].state_dict(),
                'epoch': epoch + 1,
                'mAP': mAP,
                'optimizer': self._optims[name].state_dict(),
                'scheduler': self._scheds[name].state_dict() if self._scheds[name] is not None else None,
            }, osp.join(save_dir, name), is_best=is_best)

    def set_model_mode(self, mode='train', names=None):
        assert mode in ['train', 'eval', 'test']
        for name in self.get_model_names(names):
            self._models[name].train(mode == 'train')

    def get_current_lr(self, names=None):
        return self._optims[self.get_model_names(names)[0]].param_groups[-1]['lr']

    def update_lr(self, names=None):
        for name in self.get_model_names(names):
            if self._scheds[name] is not None:
                self._scheds[name].step()

    def run(self, save_dir='log', max_epoch=0, start_epoch=0, print_freq=10, fixbase_epoch=0, open_layers=None,
            start_eval=0, eval_freq=-1, test_only=False, dist_metric='euclidean', normalize_feature=False,
            visrank=False, visrank_topk=10, use_metric_cuhk03=False, ranks=[1, 5, 10, 20], rerank=False):
        """engine.py:126-232.  Like the reference there is no evaluation / checkpoint after the last
        epoch (the `(epoch+1) != max_epoch` guard, :216)."""
        if visrank:
            raise NotImplementedError('visrank (cv2 visualisation) is out of scope')
        if rerank:
            raise NotImplementedError('re-ranking is a "next" row (SURVEY.md §8f N3)')
        if test_only:
            self.test(dist_metric=dist_metric, normalize_feature=normalize_feature, save_dir=save_dir,
                      use_metric_cuhk03=use_metric_cuhk03, ranks=ranks)
            return
        time_start = time.time()
        self.start_epoch = start_epoch
        self.max_epoch = max_epoch
        print('=> Start training')
        train_begin = time.time()
        for self.epoch in range(self.start_epoch, self.max_epoch):
            epoch_begin = time.time()
            self.train(print_freq=print_freq, fixbase_epoch=fixbase_epoch, open_layers=open_layers)
            print("Epoch Time: {}\t Total Time: {}\n".format(
                str(datetime.timedelta(seconds=int(time.time() - epoch_begin))),
                str(datetime.timedelta(seconds=int(time.time() - train_begin)))))
            if (self.epoch + 1) >= start_eval and eval_freq > 0 and (self.epoch + 1) % eval_freq == 0 \
                    and (self.epoch + 1) != self.max_epoch:
                mAP = self.test(dist_metric=dist_metric, normalize_feature=normalize_feature, save_dir=save_dir,
                                use_metric_cuhk03=use_metric_cuhk03, ranks=ranks)
                self.save_model(self.epoch, mAP, save_dir)
        print('Elapsed {}'.format(str(datetime.timedelta(seconds=round(time.time() - time_start)))))

    def train(self, print_freq=10, fixbase_epoch=0, open_layers=None):
        losses = MetricMeter()
        batch_time = AverageMeter()
        data_time = AverageMeter()
        self.set_model_mode('train')
        self.two_stepped_transfer_learning(self.epoch, fixbase_epoch, open_layers)
        self.num_batches = len(self.train_loader)
        end = time.time()
        for self.batch_idx, data in enumerate(self.train_loader):
            data_time.update(time.time() - end)
            loss_summary = self.forward_backward(data)
            batch_time.update(time.time() - end)
            losses.update(loss_summary)
            if (self.batch_idx + 1) % print_freq == 0 and ddp.rank() == 0:
                print('epoch: [{0}/{1}][{2}/{3}]\tlr {lr:.6f}\n{losses}\t'.format(
                    self.epoch + 1, self.max_epoch, self.batch_idx + 1, self.num_batches,
                    lr=self.get_current_lr(), losses=losses))
            end = time.time()
        self.update_lr()

    def forward_backward(self, data):
        raise NotImplementedError

    def test(self, dist_metric='euclidean', normalize_feature=False, visrank=False, visrank_topk=10, save_dir='',
             use_metric_cuhk03=False, ranks=[1, 5, 10, 20], rerank=False):
        self.set_model_mode('eval')
        mAP = 0.0
        for name in list(self.test_loader.keys()):
            domain = 'source' if name in self.datamanager.sources else 'target'
            print('##### Evaluating {} ({}) #####'.format(name, domain))
            rank1, mAP = self._evaluate(dataset_name=name, query_loader=self.test_loader[name]['query'],
                                        gallery_loader=self.test_loader[name]['gallery'], dist_metric=dist_metric,
                                        normalize_feature=normalize_feature, use_metric_cuhk03=use_metric_cuhk03,
                                        ranks=ranks, rerank=rerank)
        return mAP

    @torch.no_grad()
    def _evaluate(self, dataset_name='', query_loader=None, gallery_loader=None, dist_metric='euclidean',
                  normalize_feature=False, visrank=False, visrank_topk=10, save_dir='', use_metric_cuhk03=False,
                  ranks=[1, 5, 10, 20], rerank=False):
        batch_time = AverageMeter()

        def _feature_extraction(data_loader):
            f_, pids_, camids_ = [], [], []
            for data in data_loader:
                imgs, pids, camids, timeids = self.parse_data_for_eval(data)
                if self.use_gpu:
                    imgs = [im.cuda(non_blocking=True) for im in imgs]
                end = time.time()
                features = self.extract_features(imgs, timeids)
                batch_time.update(time.time() - end)
                f_.append(features.clone())             # stays on the device (reference: .cpu(), :368)
                pids_.extend(np.asarray(pids).tolist())
                camids_.extend(np.asarray(camids).tolist())
            return torch.cat(f_, 0), np.asarray(pids_), np.asarray(camids_)

        print('Extracting features from query set ...')
        qf, q_pids, q_camids = _feature_extraction(query_loader)
        print('Done, obtained {}-by-{} matrix'.format(qf.size(0), qf.size(1)))
        print('Extracting features from gallery set ...')
        gf, g_pids, g_camids = _feature_extraction(gallery_loader)
        print('Done, obtained {}-by-{} matrix'.format(gf.size(0), gf.size(1)))
        print('Speed: {:.4f} sec/batch'.format(batch_time.avg))
        if normalize_feature:
            print('Normalzing features with L2 norm ...')
            qf = F.normalize(qf, p=2, dim=1)
            gf = F.normalize(gf, p=2, dim=1)
        print('Computing distance matrix with metric={} ...'.format(dist_metric))
        if rerank:
            # reference engine.py:402-406: re-rank with the query-query and gallery-gallery matrices, then evaluate
            print('Applying person re-ranking ...')
            from .rerank import re_ranking
            distmat = compute_distance_matrix(qf, gf, dist_metric)
            distmat_qq = compute_distance_matrix(qf, qf, dist_metric)
            distmat_gg = compute_distance_matrix(gf, gf, dist_metric)
            distmat = re_ranking(distmat, distmat_qq, distmat_gg)
            print('Computing CMC and mAP for {}'.format(dataset_name))
            cmc, mAP = evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, use_metric_cuhk03=use_metric_cuhk03)
        elif ddp.world_size() > 1 and not use_metric_cuhk03:
            # every rank holds the full feature sets; each ranks its slice of the queries (ieee_amd/dist.py)
            print('Computing CMC and mAP for {} (queries sharded over {} ranks)'.format(dataset_name, ddp.world_size()))
            cmc, mAP = ddp.sharded_evaluate_rank(qf, gf, q_pids, g_pids, q_camids, g_camids, metric=dist_metric)
        else:
            distmat = compute_distance_matrix(qf, gf, dist_metric)
            print('Computing CMC and mAP for {}'.format(dataset_name))
            cmc, mAP = evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, use_metric_cuhk03=use_metric_cuhk03)
        print('** Results **')
        print('mAP: {:.2%}'.format(mAP))
        print('CMC curve')
        for r in ranks:
            if r - 1 < len(cmc):
                print('Rank-{:<3}: {:.2%}'.format(r, cmc[r - 1]))
        print('\n')
        return cmc[0], mAP

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
        """engine.py:507-529; the shipped config has fixbase_epoch=0, i.e. all layers open."""
        model = self.model if model is None else model
        if model is None:
            return
        if (epoch + 1) <= fixbase_epoch and open_layers is not None:
            if isinstance(open_layers, str):
                open_layers = [open_layers]
            print('* Only train {} (epoch: {}/{})'.format(open_layers, epoch + 1, fixbase_epoch))
            for name, module in model.named_children():
                for p in module.parameters():
                    p.requires_grad = name in open_layers
        else:
            for p in model.parameters():
                p.requires_grad = True


def _is_native(model):
    return hasattr(model, "native_net")


class _FusedStepMixin(object):
    """the native train step shared by the 3M and the CE-only engines"""

    def _fused_ok(self):
        return _is_native(self.model) and isinstance(self.optimizer, (FusedSGD, FusedAdam)) and self.use_gpu

    def _fused_step(self, imgs, pids, weight_x, weight_m, margin, eps):
        lib = _lib.require_gpu()
        m = self.model
        B, _, H, W = imgs[0].shape
        dev = m._flat_params.device
        net = m.native_net(B, H, W)
        pids = pids.to(device=dev, dtype=torch.int64).contiguous()
        m._bump_counters()
        logits, feats = net.forward(imgs, training=True)
        C = logits.shape[2]
        if not hasattr(self, "_scratch") or self._scratch[0].shape != logits.shape:
            self._scratch = (torch.empty_like(logits), torch.empty_like(feats),
                             torch.empty(18 * 2 + 3, dtype=torch.float32, device=dev),
                             torch.empty(18 * B * 2, dtype=torch.float32, device=dev),
                             torch.empty(B + 3, dtype=torch.float32, device=dev))
        dl, df, small, work, mwork = self._scratch
        head_loss, head_acc, out3 = small[:18], small[18:36], small[36:39]
        _lib.check(lib.ieee_ce_ls_fwd_bwd(_lib.ptr(logits), _lib.ptr(pids), _lib.ptr(dl), _lib.ptr(head_loss),
                                          _lib.ptr(head_acc), _lib.ptr(work), 18, B, C, float(eps),
                                          float(weight_x) * ddp.ce_grad_scale(), _lib.stream()))
        if weight_m > 0:
            _lib.check(lib.ieee_margin3m_fwd_bwd(_lib.ptr(feats), _lib.ptr(pids), _lib.ptr(df), _lib.ptr(out3),
                                                 _lib.ptr(mwork), B, feats.shape[2], float(margin), float(weight_m),
                                                 _lib.stream()))
        else:
            df.zero_()
            out3.zero_()
        staged = ddp.world_size() > 1 or (os.environ.get("IEEE_FORCE_DP_PATH") == "1" and torch.distributed.is_initialized())
        if not staged and isinstance(self.optimizer, FusedSGD) and os.environ.get("IEEE_OPT_OVERLAP", "1") != "0":
            # single GPU: parts 0-3 (head, layer4, layer3, layer2 = 90 % of the parameters) are updated on a helper
            # stream while the compute stream still runs the backward of layer1 + stem and the side stream drains the
            # last weight gradients; only layer1 + stem wait for the final join.  Same arithmetic as optimizer.step().
            main = torch.cuda.current_stream()
            if not hasattr(self, "_opt_stream"):
                self._opt_stream = torch.cuda.Stream()
            helper = self._opt_stream
            for part in range(4):
                net.backward_part_async(dl, df, part)
            helper.wait_stream(main)
            net.side_wait(helper)
            net.backward_part_async(dl, df, 4)
            with torch.cuda.stream(helper):
                for part in range(4):
                    self.optimizer.step_part(part)
            net.side_wait()
            self.optimizer.step_part(4)
            main.wait_stream(helper)
            return small, out3
        if not staged:
            net.backward(dl, df)
        else:
            # data parallel: the backward runs in 5 parts; as soon as a part is done its (final) slice of the flat
            # gradient is all-reduced (sum) asynchronously over RCCL/xGMI while the next part computes.  Together
            # the slices are exactly one pass over the 438 MB buffer; torch orders each collective after the
            # kernels already enqueued on the compute stream, and wait() orders the SGD step after them.
            # The compute stream is never blocked between parts: a helper stream waits for the part's kernels on the
            # compute stream AND for its weight gradients on the executor's side stream, and the collective is issued
            # from there.
            handles = []
            main = torch.cuda.current_stream()
            if not hasattr(self, "_comm_stream"):
                self._comm_stream = torch.cuda.Stream()
            comm = self._comm_stream
            for part, ranges in enumerate(m.grad_part_ranges()):
                net.backward_part_async(dl, df, part)
                comm.wait_stream(main)
                net.side_wait(comm)
                with torch.cuda.stream(comm):
                    for a, b in ranges:
                        handles.append(torch.distributed.all_reduce(m._flat_grads[a:b], op=torch.distributed.ReduceOp.SUM,
                                                                    async_op=True))
            net.side_wait()                      # final join of the weight-gradient stream into the compute stream
            for h in handles:
                h.wait()
            main.wait_stream(comm)
        self.optimizer.step()
        return small, out3

    def _summary_from(self, small):
        # the step's single device->host read-back, into a pinned buffer (a pageable .cpu() adds ~50 us of staging)
        host = getattr(self, "_small_host", None)
        if host is None or host.numel() != small.numel():
            host = self._small_host = torch.empty(small.numel(), dtype=small.dtype, pin_memory=True)
        host.copy_(small, non_blocking=True)
        torch.cuda.current_stream(small.device).synchronize()
        v = host.numpy().copy()
        hl, ha = v[:18], v[18:36]
        lR, lN, lT = float(hl[0:6].sum()), float(hl[6:12].sum()), float(hl[12:18].sum())
        aR, aN, aT = float(ha[0:6].mean()), float(ha[6:12].mean()), float(ha[12:18].mean())
        return lR, lN, lT, aR, aN, aT, float(v[36]), v[37], v[38]


class Image3MEngine(_FusedStepMixin, Engine):
    """CE (x18, label smoothed) + 3M margin loss engine, reference engine/image/margin.py:62-154."""

    def __init__(self, datamanager, model, optimizer, margin=3, weight_m=1, weight_x=1, scheduler=None, use_gpu=True,
                 label_smooth=True):
        super(Image3MEngine, self).__init__(datamanager, use_gpu)
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.register_model('model', model, optimizer, scheduler)
        assert weight_m >= 0 and weight_x >= 0
        assert weight_m + weight_x > 0
        self.weight_m = weight_m
        self.weight_x = weight_x
        self.margin = margin
        self.criterion_m = multiModalMarginLossNew(margin=margin)
        self.criterion_x = CrossEntropyLoss(num_classes=self.datamanager.num_train_pids, use_gpu=self.use_gpu,
                                            label_smooth=label_smooth)

    def forward_backward(self, data):
        imgs, pids, timeids = self.parse_data_for_train(data)
        if self.use_gpu:
            imgs = [im.cuda(non_blocking=True) for im in imgs]
            pids = pids.cuda()
        if self._fused_ok():
            small, out3 = self._fused_step(imgs, pids, self.weight_x, self.weight_m, self.margin, self.criterion_x.eps)
            lR, lN, lT, aR, aN, aT, lm, n_id, n_chunk = self._summary_from(small)
            if self.weight_m > 0 and n_chunk < n_id:
                raise IndexError('tuple index out of range')    # what chunk()/feat1[i] raises in the reference
            loss_x = lR + lN + lT
            return {'loss': self.weight_m * lm + self.weight_x * loss_x, 'LossX': loss_x, 'LossM': out3[0].clone(),
                    'accR': aR, 'lossR': lR, 'accN': aN, 'lossN': lN, 'accT': aT, 'lossT': lT}
        # ---- generic path: same structure as the reference step (margin.py:102-152)
        outputs_R, outputs_N, outputs_T, features_RGB, features_NI, features_TI = self.model(imgs)
        loss = 0
        loss_m = 0
        if self.weight_m > 0:
            loss_m = self.criterion_m(features_RGB, features_NI, features_TI, pids)
            loss += self.weight_m * loss_m
        if self.weight_x > 0:
            loss_R = self.compute_loss(self.criterion_x, outputs_R, pids)
            loss_N = self.compute_loss(self.criterion_x, outputs_N, pids)
            loss_T = self.compute_loss(self.criterion_x, outputs_T, pids)
            loss_x = loss_R + loss_N + loss_T
            loss += self.weight_x * loss_x
        self.optimizer.zero_grad()
        loss.backward()
        if ddp.world_size() > 1:
            raise RuntimeError("multi-GPU training uses the fused path (FusedSGD); see ieee_amd.dist")
        self.optimizer.step()
        acc_R = acc_N = acc_T = 0
        for i in range(len(outputs_R)):
            acc_R += accuracy(outputs_R[i], pids)[0]
            acc_N += accuracy(outputs_N[i], pids)[0]
            acc_T += accuracy(outputs_T[i], pids)[0]
        acc_R /= len(outputs_R)
        acc_N /= len(outputs_N)
        acc_T /= len(outputs_T)
        return {'loss': loss.item(), 'LossX': loss_x.item(), 'LossM': loss_m, 'accR': acc_R.item(),
                'lossR': loss_R.item(), 'accN': acc_N.item(), 'lossN': loss_N.item(), 'accT': acc_T.item(),
                'lossT': loss_T.item()}


class MultiModalImageSoftmaxEngine(_FusedStepMixin, Engine):
    """CE-only 3-modal engine (the "3M off" ablation), reference engine/image/softmax.py:11-132."""

    def __init__(self, datamanager, model, optimizer, scheduler=None, use_gpu=True, label_smooth=True):
        super(MultiModalImageSoftmaxEngine, self).__init__(datamanager, use_gpu)
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.register_model('model', model, optimizer, scheduler)
        self.criterion = CrossEntropyLoss(num_classes=self.datamanager.num_train_pids, use_gpu=self.use_gpu,
                                          label_smooth=label_smooth)

    def forward_backward(self, data):
        imgs, pids, timeids = self.parse_data_for_train(data)
        if self.use_gpu:
            imgs = [im.cuda(non_blocking=True) for im in imgs]
            pids = pids.cuda()
        if self._fused_ok():
            small, _ = self._fused_step(imgs, pids, 1.0, 0.0, 0.0, self.criterion.eps)
            lR, lN, lT, aR, aN, aT = self._summary_from(small)[:6]
            return {'loss_all': lR + lN + lT, 'loss_R': lR, 'acc_R': aR, 'loss_N': lN, 'acc_N': aN, 'loss_T': lT,
                    'acc_T': aT}
        out = self.model(imgs)
        outputs_R, outputs_N, outputs_T = out[0], out[1], out[2]
        loss_R = self.compute_loss(self.criterion, outputs_R, pids)
        loss_N = self.compute_loss(self.criterion, outputs_N, pids)
        loss_T = self.compute_loss(self.criterion, outputs_T, pids)
        loss = loss_R + loss_N + loss_T
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        acc = [sum(accuracy(o, pids)[0] for o in oo) / len(oo) for oo in (outputs_R, outputs_N, outputs_T)]
        return {'loss_all': loss.item(), 'loss_R': loss_R.item(), 'acc_R': acc[0].item(), 'loss_N': loss_N.item(),
                'acc_N': acc[1].item(), 'loss_T': loss_T.item(), 'acc_T': acc[2].item()}
