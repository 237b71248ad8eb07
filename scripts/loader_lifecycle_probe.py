"""where does a tiny loader spend its wall time on the GPU box? (build, first batch, epoch, partial epoch, deletion)"""
import gc, os, sys, tempfile, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from ieee_amd import data as D
torch.zeros(1).cuda()
tmp = tempfile.mkdtemp()
rng = np.random.RandomState(2)
names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + k % 4, k) for pid in (3, 9, 20, 31) for k in range(4)]
for split in ("train_171", "test"):
    for mod in ("RGB", "NI", "TI"):
        d = os.path.join(tmp, "RGBNT201", split, mod); os.makedirs(d)
        for n in names:
            Image.fromarray(rng.randint(0, 256, size=(64, 32, 3)).astype(np.uint8), "RGB").save(os.path.join(d, n), quality=92)
ds = D.RGBNT201(root=tmp)
held = []
for prefetch, gb in ((2, 0), (2, 20), (2, 60), (0, 60)):
    while len(held) * 10 < gb:
        held.append(torch.empty(10 << 30, dtype=torch.uint8, device="cuda").zero_())     # touched VRAM held by the process
    torch.cuda.synchronize()
    t0 = time.time(); pid = os.fork()
    if pid == 0:
        os._exit(0)
    os.waitpid(pid, 0)
    print("with %d GB of device memory held: a bare fork + exit takes %.2f s" % (gb, time.time() - t0), flush=True)
    t = [time.time()]
    train, query, gallery = D.build_loaders(ds, 256, 128, "random_flip", batch_size_train=8, batch_size_test=5, workers=2, prefetch=prefetch)
    t.append(time.time())
    it = iter(train); b = next(it); torch.cuda.synchronize(); t.append(time.time())
    del it, b; gc.collect(); t.append(time.time())
    n = sum(1 for _ in train); torch.cuda.synchronize(); t.append(time.time())
    q = next(iter(query)); t.append(time.time())
    del q; gc.collect(); t.append(time.time())
    del train; gc.collect(); t.append(time.time())
    del query, gallery; gc.collect(); t.append(time.time())
    print("prefetch %d: build %.2f | first batch %.2f | drop iterator %.2f | epoch (%d) %.2f | first query batch %.2f | drop it %.2f | "
          "del train %.2f | del query+gallery %.2f" % ((prefetch,) + tuple(t[i + 1] - t[i] for i in range(3)) + (n,) + tuple(t[i + 1] - t[i] for i in range(3, 8))), flush=True)
