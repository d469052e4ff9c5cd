#!/bin/bash
# r07lc: a lone frame set's H2D copies on stream A itself (OATGPU_LONE_COPY_A=1, measurement build preloaded into oat-track-hip) against the
# copy streams: frame posted -> position token at 1080p through the process pipeline
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from oat_amd.synth import SyntheticStream
st = SyntheticStream(1080, 1920, 0, n_discs=2)
np.stack([st.frame(9 * t, with_discs=t > 0) for t in range(8)]).tofile("/dev/shm/r07lc.raw")
PY
B=$R/build/bin
{
for rep in 1 2 3; do for a in 0 1; do for rate in "" "-r 500"; do
  $B/oat-clean-hip lc_cam lc_trk > /dev/null 2>&1
  $B/oat-latency-probe lc_cam lc_trk -f /dev/shm/r07lc.raw --rows 1080 --cols 1920 -n 1000 $rate > $O/lc_probe.out 2> /dev/null &
  pp=$!
  sleep 0.5
  OATGPU_LONE_COPY_A=$a LD_PRELOAD=$R/build/variants/liboatgpu_meas.so $B/oat-track-hip lc_cam lc_trk -a 0.01 --area "[20,100000]" -H "[100,125]" -S "[150,256]" -V "[100,256]" -e 3 -d 7 --ring 2 --gpu-index 0 > /dev/null 2>&1
  wait $pp
  echo "copies on ${a/0/the copy streams}${a/1/} $( [ $a = 1 ] && echo stream A ) ${rate:-free-running}: $(grep '^{' $O/lc_probe.out | tail -1 | cut -c1-200)"
done; done; done
} 2>&1 | tee $O/r07lc_lone_copy_on_a.txt
rm -f /dev/shm/r07lc.raw
