import os, sys, torch
sys.path.insert(0, os.getcwd())
from footprints_amd import ops, _lib as L
N,H,W=12,192,640
img=torch.rand(N,3,H,W,device="cuda"); g=torch.randn(N,H//2,W//2,64,device="cuda"); dw=torch.empty(64,3,7,7,device="cuda")
d=ops.make_desc(N,H//2,W//2,H,W,3,0,64,7,2,3,L.GATHER_STEM)
for _ in range(3): ops.conv_wgrad(d,img,None,g,dw)
torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.conv_wgrad(d,img,None,g,dw)
e.record(); torch.cuda.synchronize()
print("stem wgrad %.1f us" % (s.elapsed_time(e)/20*1e3))
