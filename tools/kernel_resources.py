"""Register / scratch / occupancy table of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py minsu3d_amd/csrc/spconv.hip > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    cmd = ["hipcc", "-c", os.path.join(root, src), "-o", os.path.join(d, "o.o"), "--offload-arch=gfx950", "-O3", "-std=c++17",
           "-fPIC", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(root, "include")]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
print(f"kernel resource usage of {src} (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)")
print(f"{'kernel':<72}{'VGPR':>6}{'AGPR':>6}{'scratch B/lane':>16}{'waves/SIMD':>12}{'LDS B':>8}")
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" [")[0].strip()
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except FileNotFoundError:
        dem = name
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)
    dem = re.sub(r"^void ", "", dem).split("(")[0]
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    v, a, sc, oc, ld = g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dem[:70]:<72}{v:>6}{a:>6}{sc:>16}{oc:>12}{ld:>8}")
