#!/bin/bash
# round 4 experiment: k_wt_gru2 (one wave per SIMD, two row tiles per wave) against k_wt_gru on the launch-per-sub-step path
mkdir -p gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TMPDIR=/tmp
cat > /tmp/dump.py <<'PY'
import sys, hashlib, json
sys.path.insert(0, '.')
import numpy as np
import bench
args = bench.parse(['--config', '2', '--utterances', '64', '--frames', '200', '--no_cpu_baseline', '--no_extra_configs'])
import torch
from uisrnn_amd import _capi
w = bench.Workload(dict(bench.CONFIGS[2]), args, 0, 1, torch.device('cuda:0'), 0)
out = w.decoder.decode_device(w.d_frames.data_ptr(), w.offsets, w.beam, w.look, w.tau, w.d_labels.data_ptr(), w.d_scores.data_ptr(),
                              max_clusters=12, flags=_capi.UIS_FLAG_STEPWISE | _capi.UIS_FLAG_PROFILE)
torch.cuda.synchronize()
lab = w.d_labels.cpu().numpy(); sc = w.d_scores.cpu().numpy()
print(json.dumps({'labels': hashlib.sha1(lab.tobytes()).hexdigest(), 'scores': hashlib.sha1(sc.tobytes()).hexdigest(),
                  'kernel': out['stats']['decode_kernel'], 'ms': {k: round(v, 3) for k, v in out['stats']['kernel_ms'].items()}}))
PY
for env in "UIS_X=0" "UIS_WT_GRU2=1" "UIS_X=0" "UIS_WT_GRU2=1"; do
  echo "== $env" | tee -a gpurun_out/r04t_gru2.txt
  env $env python /tmp/dump.py 2>&1 | tail -1 | tee -a gpurun_out/r04t_gru2.txt
done
