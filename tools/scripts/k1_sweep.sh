cd "${GRAFT_REPO_ROOT:?}" || exit 1
for c in 0 1024 2048 4096 8192; do
  MS3D_WGRAD_K1_CHUNKS=$c python3 - <<'PY'
import os, torch, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from minsu3d_amd import backend as B
be = B.get_backend()
n = 575000
for cin, cout in ((16, 16), (16, 20), (16, 3)):
    x = torch.randn(n, cin, device="cuda"); g = torch.randn(n, cout, device="cuda")
    ident = be.identity_table(n, x.device)
    for _ in range(3): be.conv_backward_weight(x, g, ident, n, 1, cin, cout)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): be.conv_backward_weight(x, g, ident, n, 1, cin, cout)
    e1.record(); torch.cuda.synchronize()
    print(os.environ["MS3D_WGRAD_K1_CHUNKS"], cin, cout, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
PY
done
