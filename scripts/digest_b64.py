"""one-off: digest of the flat gradient buffer after a B = 64 bf16 forward + backward (x3 repeats), to compare scheduling switches"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import make_model, images, C
B = 64
m = make_model(5, dtype=torch.bfloat16).train()
xs = [x.cuda() for x in images(B, 5)]
net = m.native_net(B, 256, 128)
g = torch.Generator(device="cuda").manual_seed(3)
dl = torch.randn(18, B, C, generator=g, device="cuda") * 1e-2
df = torch.randn(3, B, 768, generator=g, device="cuda") * 1e-2
m._bump_counters()
for rep in range(3):
    m._flat_grads.zero_()
    net.forward(xs, training=True)
    net.backward(dl, df)
    torch.cuda.synchronize()
    print("DIGEST", hashlib.sha1(m._flat_grads.cpu().numpy().tobytes()).hexdigest())
