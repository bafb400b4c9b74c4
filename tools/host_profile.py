"""cProfile of the host side of train steps (where the Python time goes)."""
import sys, os, time, torch, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from minsu3d_amd.config import load_config
cfg = load_config(); dev = torch.device("cuda", 0)
model = bench.build(cfg, dev); opt = model.configure_optimizers()
batches = [bench.make_batch([4 * i + j for j in range(4)], dev) for i in range(4)]
for i in range(5): bench.train_step(model, model, opt, batches[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10): bench.train_step(model, model, opt, batches[i % 4])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
