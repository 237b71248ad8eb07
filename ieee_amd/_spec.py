"""state_dict specification of IEEE3modalPart: every key, shape and kind, in the reference's
registration order (reference torchreid/models/ieee3modalPart.py:286-393 and
torchreid/models/resnet.py:135-161, 443-571).  Written from the architecture,
and pinned against the imported reference's state_dict by tests/golden/state_dict_spec.txt."""

MODAL = 3
PARTS = 6


def _bn(prefix, c):
    return [(prefix + ".weight", (c,), "param"), (prefix + ".bias", (c,), "param"),
            (prefix + ".running_mean", (c,), "buffer"), (prefix + ".running_var", (c,), "buffer"),
            (prefix + ".num_batches_tracked", (), "counter")]


def _conv(name, co, ci, k):
    return [(name + ".weight", (co, ci, k, k), "param")]


def _linear(name, out_f, in_f):
    return [(name + ".weight", (out_f, in_f), "param"), (name + ".bias", (out_f,), "param")]


def _resnet50(prefix):
    """ResNetIEEE with layers [3,4,6,3], Bottleneck, last_stride=1 (resnet.py:1248-1256)."""
    out = _conv(prefix + "conv1", 64, 3, 7) + _bn(prefix + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for b in range(blocks):
            p = "%slayer%d.%d." % (prefix, li + 1, b)
            out += _conv(p + "conv1", planes, inplanes, 1) + _bn(p + "bn1", planes)
            out += _conv(p + "conv2", planes, planes, 3) + _bn(p + "bn2", planes)
            out += _conv(p + "conv3", planes * 4, planes, 1) + _bn(p + "bn3", planes * 4)
            if b == 0:   # stride != 1 or inplanes != planes*4 -> downsample (resnet.py:546-550)
                out += _conv(p + "downsample.0", planes * 4, inplanes, 1) + _bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    return out


def state_spec(num_classes, interaction=True, attention=True, using_rem=True):
    """[(key, shape, kind)] with kind in {param, buffer, counter}."""
    s = []
    for m in range(MODAL):
        s += _resnet50("backbone.%d." % m)
    if interaction:
        for name in ("convOne", "convAvgRest"):
            for m in range(MODAL):
                s += _conv("%s.%d.layers.0" % (name, m), 2048, 2048, 1) + _bn("%s.%d.layers.1" % (name, m), 2048)
        if attention:
            for m in range(MODAL):
                s += _conv("CA.%d.fc.0" % m, 128, 2048, 1) + _conv("CA.%d.fc.2" % m, 2048, 128, 1)
    for m in range(MODAL):
        s += _conv("reduce_layer.%d.layers.0" % m, 768, 2048, 1) + _bn("reduce_layer.%d.layers.1" % m, 768)
    if using_rem:
        for m in range(MODAL):
            p = "REM.%d." % m
            s += _linear(p + "conv_query", 768, 768) + _linear(p + "conv_part", 768, 768) \
                + _linear(p + "conv_value", 768, 768)
            s.insert(len(s) - 6, (p + "param", (1,), "param"))
    for letter in ("R", "T", "N"):          # fc_T is registered before fc_N (ieee3modalPart.py:354-371)
        for i in range(PARTS):
            p = "fc_%s.%d." % (letter, i)
            s += _linear(p + "0", 128, 768) + _bn(p + "1", 128)
    for letter in ("R", "N", "T"):          # classifiers in R, N, T order (:374-391)
        for i in range(PARTS):
            s += _linear("classifier_%s.%d" % (letter, i), num_classes, 128)
    return s
