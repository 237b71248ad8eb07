"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the
IEEE3modalPart train / eval step in stock torch fp32 ops, arranged as the reference arranges them.

  bottleneck / resnet50_trunk   <- torchreid/models/resnet.py:164-184, 496-523, 622-635
  cim / channel_attention       <- torchreid/models/ieee3modalPart.py:266-282, 427-435
  forward = trunks + tail       <- torchreid/models/ieee3modalPart.py:439-523
  rem (closed form)             <- torchreid/models/ieee3modalPart.py:60-80
  cross_entropy_ls              <- torchreid/losses/cross_entropy_loss.py:36-50
  margin3m                      <- torchreid/losses/multi_modal_margin_loss_new.py:19-40
  train_step                    <- torchreid/engine/image/margin.py:94-154 (engine="margin") or
                                   engine/image/softmax.py:81-132 (engine="softmax") + optim/optimizer.py:130-138;
                                   frozen=... <- engine/engine.py:507-529 + utils/torchtools.py:183-221

Works on a plain dict name -> tensor with the reference's state_dict keys.  Pinned against the imported
reference by tests/test_model_oracle.py (here) and tests/golden/model_golden.npz / model_golden_r2.npz (everywhere).
"""
import torch
import torch.nn.functional as F

EPS, MOM = 1e-5, 0.1
# top-level children in eval() mode while the rest trains (open_specified_layers, torchreid/utils/torchtools.py:183-221);
# set through train_step(frozen=...)
FROZEN = ()


def _live(p, training):
    """train-mode statistics for the BatchNorm at state-dict prefix p?  (False inside a frozen child)"""
    return bool(training) and p.split(".")[0] not in FROZEN


def _bn(x, sd, p, training):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        _live(p, training), MOM, EPS)


def bottleneck(x, sd, p, stride, has_ds, training):
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1", training))
    out = F.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], None, stride, 1), sd, p + "bn2", training))
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3", training)
    identity = x
    if has_ds:
        identity = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride), sd, p + "downsample.1", training)
    return F.relu(out + identity)


def resnet50_trunk(x, sd, p, training, taps=None):
    x = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3), sd, p + "bn1", training))
    if taps is not None:
        taps[p + "stem"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in enumerate(zip((3, 4, 6, 3), (1, 2, 2, 1))):     # last_stride = 1
        for b in range(blocks):
            x = bottleneck(x, sd, "%slayer%d.%d." % (p, li + 1, b), stride if b == 0 else 1, b == 0, training)
        if taps is not None:
            taps["%slayer%d" % (p, li + 1)] = x
    return x


def channel_attention(x, sd, p):
    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, sd[p + "fc.0.weight"])), sd[p + "fc.2.weight"])
    return torch.sigmoid(mlp(F.adaptive_avg_pool2d(x, 1)) + mlp(F.adaptive_max_pool2d(x, 1)))


def dim_reduce(x, sd, p, training):
    return F.relu(_bn(F.conv2d(x, sd[p + "layers.0.weight"]), sd, p + "layers.1", training))


def forward(sd, xs, training, loss="margin", interaction=True, attention=True, using_rem=True, taps=None):
    """xs = [RGB, NI, TI]; sd's running stats are updated in place when training (pass clones)."""
    f = [resnet50_trunk(xs[m], sd, "backbone.%d." % m, training, taps) for m in range(3)]
    return tail(sd, f, training, loss, interaction, attention, using_rem, taps)


def tail(sd, f, training, loss="margin", interaction=True, attention=True, using_rem=True, taps=None, conv_out=None):
    """everything behind the three trunks (ieee3modalPart.py:445-523): f = the three [B, 2048, 16, 8] trunk maps.  Split
    out of forward() so that a test can hand it the trunk maps of another implementation and differentiate from there.
    conv_out = {"one": [3 maps], "rest": [3 maps]} (optional): the OUTPUTS of the convOne / convAvgRest convolutions, used
    instead of computing them -- a test of a reduced-precision implementation starts behind its rounded conv outputs, so
    that the ReLU masks of both sides are taken from the same numbers."""
    pooled, glob = [], []
    if interaction:
        for m in range(3):
            a, b = [k for k in range(3) if k != m]
            if conv_out is not None:
                one = F.relu(_bn(conv_out["one"][m], sd, "convOne.%d.layers.1" % m, training))
                rest = F.relu(_bn(conv_out["rest"][m], sd, "convAvgRest.%d.layers.1" % m, training))
            else:
                one = dim_reduce(f[m], sd, "convOne.%d." % m, training)
                rest = dim_reduce(f[a] + f[b], sd, "convAvgRest.%d." % m, training)
            if attention:
                rest = channel_attention(rest, sd, "CA.%d." % m) * rest + rest
            pooled.append(one + rest)
    else:
        pooled = list(f)
    for m in range(3):      # reduce_layer: global vector first, then the 6 parts (two BN calls, :449-455)
        glob.append(dim_reduce(F.adaptive_avg_pool2d(f[m], (1, 1)), sd, "reduce_layer.%d." % m, training))
    for m in range(3):
        pooled[m] = dim_reduce(F.adaptive_avg_pool2d(pooled[m], (6, 1)), sd, "reduce_layer.%d." % m, training)
    glob = [g.flatten(1) for g in glob]
    parts = [[pooled[m][:, :, i, 0] for i in range(6)] for m in range(3)]
    if taps is not None:
        taps["glob"] = torch.stack(glob)
        taps["parts_pre_rem"] = torch.stack([torch.stack(p, 1) for p in parts])
    if using_rem:
        for m in range(3):
            r = F.linear(glob[m], sd["REM.%d.conv_part.weight" % m], sd["REM.%d.conv_part.bias" % m])
            parts[m] = [q + 2.0 * sd["REM.%d.param" % m] * r for q in parts[m]]
    letters = "RNT"
    fc = [[F.relu(F.batch_norm(F.linear(parts[m][i], sd["fc_%s.%d.0.weight" % (letters[m], i)],
                                        sd["fc_%s.%d.0.bias" % (letters[m], i)]),
                               sd["fc_%s.%d.1.running_mean" % (letters[m], i)],
                               sd["fc_%s.%d.1.running_var" % (letters[m], i)],
                               sd["fc_%s.%d.1.weight" % (letters[m], i)], sd["fc_%s.%d.1.bias" % (letters[m], i)],
                               _live("fc_%s" % letters[m], training), MOM, EPS)) for i in range(6)] for m in range(3)]
    cat = [torch.cat(fc[m], 1) for m in range(3)]
    if not training:
        return torch.cat([cat[2], cat[0], cat[1]], 1)          # T, R, N   (:502)
    logits = [[F.linear(fc[m][i], sd["classifier_%s.%d.weight" % (letters[m], i)],
                        sd["classifier_%s.%d.bias" % (letters[m], i)]) for i in range(6)] for m in range(3)]
    if loss == "softmax":
        return logits[0], logits[1], logits[2]
    return logits[0], logits[1], logits[2], F.normalize(cat[0]), F.normalize(cat[1]), F.normalize(cat[2])


def cross_entropy_ls(logits, target, num_classes, eps=0.1):
    logp = F.log_softmax(logits, 1)
    t = torch.zeros_like(logp).scatter_(1, target.unsqueeze(1), 1)
    t = (1 - eps) * t + eps / num_classes
    return (-t * logp).mean(0).sum()


def margin3m(f1, f2, f3, pids, margin):
    n = len(pids.unique())
    c1, c2, c3 = f1.chunk(n, 0), f2.chunk(n, 0), f3.chunk(n, 0)
    total = 0
    for i in range(n):
        a, b, c = c1[i].mean(0), c2[i].mean(0), c3[i].mean(0)
        d = lambda u, v: ((u - v) ** 2).sum()
        total = total + max(abs(margin - d(a, b)), abs(margin - d(b, c)), abs(margin - d(a, c)))
    return total


def losses(outputs, pids, num_classes, margin=1.0, weight_m=1.0, weight_x=1.0):
    oR, oN, oT, fR, fN, fT = outputs
    loss_m = margin3m(fR, fN, fT, pids, margin)
    lR = sum(cross_entropy_ls(o, pids, num_classes) for o in oR)
    lN = sum(cross_entropy_ls(o, pids, num_classes) for o in oN)
    lT = sum(cross_entropy_ls(o, pids, num_classes) for o in oT)
    loss = weight_m * loss_m + weight_x * (lR + lN + lT)
    acc = [sum(100.0 * (o.argmax(1) == pids).float().mean() for o in oo) / 6 for oo in (oR, oN, oT)]
    return loss, dict(loss=loss, LossX=lR + lN + lT, LossM=loss_m, lossR=lR, lossN=lN, lossT=lT,
                      accR=acc[0], accN=acc[1], accT=acc[2])


PARAM_LEAVES = ("weight", "bias", "param")


def split_state(sd):
    """(params requiring grad, buffers) from a flat state dict"""
    params = {k: v for k, v in sd.items() if k.rsplit(".", 1)[-1] in PARAM_LEAVES}
    bufs = {k: v for k, v in sd.items() if k not in params}
    return params, bufs


def softmax_losses(outputs, pids, num_classes):
    """MultiModalImageSoftmaxEngine.forward_backward's loss and summary (engine/image/softmax.py:94-130)"""
    oR, oN, oT = outputs[:3]
    lR = sum(cross_entropy_ls(o, pids, num_classes) for o in oR)
    lN = sum(cross_entropy_ls(o, pids, num_classes) for o in oN)
    lT = sum(cross_entropy_ls(o, pids, num_classes) for o in oT)
    acc = [sum(100.0 * (o.argmax(1) == pids).float().mean() for o in oo) / 6 for oo in (oR, oN, oT)]
    loss = lR + lN + lT
    return loss, dict(loss_all=loss, loss_R=lR, acc_R=acc[0], loss_N=lN, acc_N=acc[1], loss_T=lT, acc_T=acc[2])


def calibrate_running_stats(sd, xs, **flags):
    """running statistics := batch statistics of xs (one train-mode forward with momentum 1), in place; what
    tests/golden/gen_model_golden_r2.py does to the reference model before its evaluation cases"""
    global MOM
    keep, MOM = MOM, 1.0
    try:
        with torch.no_grad():
            forward(sd, xs, True, "margin", **flags)
    finally:
        MOM = keep
    return sd


def train_step(sd, xs, pids, num_classes, lr=1e-3, momentum=0.9, wd=5e-4, mom_state=None, margin=1.0, engine="margin",
               frozen=(), **flags):
    """one engine step on CPU: returns (summary, grads, new_state, new_momentum).  SGD with nesterov
    (hard-coded in the reference, optim/optimizer.py:137), dampening 0.  engine = "margin" (Image3MEngine) or
    "softmax" (MultiModalImageSoftmaxEngine: CE only, model built with loss='softmax')."""
    global FROZEN
    sd = {k: v.clone() for k, v in sd.items()}
    params, _ = split_state(sd)
    # frozen children (two-stepped transfer learning, engine.py:507-529): eval()-mode BatchNorms, no gradient, no update
    params = {k: v for k, v in params.items() if k.split(".")[0] not in frozen}
    for p in params.values():
        p.requires_grad_(True)
    keep, FROZEN = FROZEN, tuple(frozen)
    try:
        out = forward(sd, xs, True, engine, **flags)
    finally:
        FROZEN = keep
    if engine == "softmax":
        loss, summary = softmax_losses(out, pids, num_classes)
    else:
        loss, summary = losses(out, pids, num_classes, margin)
    names = list(params)
    g = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    grads = dict(zip(names, g))
    if flags.get("using_rem", True):
        # the literal nonLocal.forward routes conv_query through softmax over ONE element, so autograd
        # hands it exact zeros (not None) and SGD still decays it; conv_value stays None
        for k in names:
            if ".conv_query." in k and grads[k] is None:
                grads[k] = torch.zeros_like(params[k])
    mom_state = dict(mom_state or {})
    with torch.no_grad():
        for k in names:
            if grads[k] is None:
                continue
            d = grads[k] + wd * params[k]
            buf = mom_state.get(k)
            buf = d.clone() if buf is None else momentum * buf + d
            mom_state[k] = buf
            params[k].sub_(lr * (d + momentum * buf))
    new_sd = {k: v.detach() for k, v in sd.items()}
    return {k: float(v.detach()) for k, v in summary.items()}, grads, new_sd, mom_state
