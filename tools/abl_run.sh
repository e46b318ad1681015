timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv1_block or conv1_1" 2>&1 | tail -5
python - <<'PY'
import torch, time, os, sys
sys.path.insert(0, '.')
from sgg_amd import ops
from tools.gemm_bench import timeit
dev='cuda:0'
B,H=8,608
img=torch.zeros(B,H+2,H+2,4,device=dev); img[:,1:-1,1:-1,:3]=torch.randn(B,H,H,3,device=dev)
w1=torch.randn(64,27,device=dev)/5; b1=torch.randn(64,device=dev)*.1
for dt in (torch.float16, torch.bfloat16):
    w2=(torch.randn(64,3,3,64,device=dev)/24).to(dt); b2=torch.randn(64,device=dev)*.1
    y1=torch.zeros(B,H+2,H+2,64,device=dev,dtype=dt); y2=torch.zeros(B,H//2+2,H//2+2,64,device=dev,dtype=dt)
    t1=timeit(lambda: ops.conv1_1(img,w1,b1,y1),reps=20); t2=timeit(lambda: ops.conv3x3_relu(y1,w2,b2,y2,1,pool=True),reps=20)
    fr=ops.conv1_pack_weights(w1,dt); t3=timeit(lambda: ops.conv1_block(img,fr,b1,w2,b2,y2,1,pool=True),reps=20)
    print(dt, 'conv1_1 %.1f us + conv1_2 %.1f us = %.1f ; fused %.1f us' % (t1*1e3,t2*1e3,(t1+t2)*1e3,t3*1e3))
PY
