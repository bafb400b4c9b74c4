import sys, os, collections, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd import backend as B
from minsu3d_amd.config import load_config
cfg = load_config(); dev = torch.device("cuda", 0)
be = B.get_backend()
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batch = bench.make_batch([0,1,2,3], dev)
class T(B.KernelTimer):
    def begin(self, name, K, cin, cout, nbr):
        ev = super().begin(name, K, cin, cout, nbr)
        if ev is not None: self.meta.append((K, cin, cout, nbr.shape[1]))
        return ev
t = T(lambda *a: True, be.lib, max_records=100000); t.meta = []
for i in range(2): bench.train_step(model, model, opt, batch)
be.kernel_timer = t; t.enabled = True
for i in range(3): bench.train_step(model, model, opt, batch)
torch.cuda.synchronize()
ms = [be.lib.ms3d_event_elapsed_ms(a, b) for a, b, _ in t.records]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (a, b, nb), m, meta in zip(t.records, ms, t.meta):
    g = agg[meta]; g[0] += 1; g[1] += m; g[2] += nb
print("K cin cout V : launches/step avg_us GB/s(algorithmic) ms/step")
tot = 0
for k, (n, m, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, f"{n/3:5.1f} {1e3*m/n:8.1f} {nb/m/1e6:8.0f} {m/3:6.2f}")
    tot += m / 3
print("total fwd conv ms/step", tot)
