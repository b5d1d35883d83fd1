cd /root/repo
for env in "HSSK_SWEEP_MMA_NC_BIG=16" "HSSK_SWEEP_MMA_NC_BIG=32" "HSSK_SWEEP_MMA_NC_BIG=64" "HSSK_SWEEP_MMA_NC_BIG=32 HSSK_SWEEP_MMA_T_BIG=512" "HSSK_SWEEP_MMA_NC=32" ; do
  echo "== $env"; env $env python tools/sweep_ab.py 100000 256 64 2>&1 | grep -v amdgpu.ids
done
