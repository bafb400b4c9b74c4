# K = 8 layers on the kernels that can serve them (bf16x3 table walk = default for wide layers, weight-stream pair list, f32 table walk)
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
for cfg in "96 64 8u 1" "128 96 8u 2" "64 96 8 1" "96 128 8 2" "64 32 8u 0" "160 128 8u 3" "48 32 8u 1" "64 48 8u 2"; do
  for env in "X=1" "MS3D_PAIRSTREAM=3 MS3D_BF16X3=0" "MS3D_BF16X3=0 MS3D_PAIRSTREAM=0"; do
    env $env python3 tools/conv_micro.py $cfg 2>&1 | grep -o "cin=[0-9]* cout=[0-9]* K=[0-9]* vin=[0-9]* vout=[0-9]*\|layer fwd [0-9.]* us\|wgrad [0-9.]* us" | tr '\n' ' '; echo " [$env]"
  done
done
