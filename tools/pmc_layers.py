import sys, torch
sys.path.insert(0, '.')
from mft_amd import ops
P,h,w=7,64,64; M=P*h*w
for name,cin,cout,kh,kw in [("gruzr",384,256,1,5),("ou1",712,256,3,3),("convc1",324,256,1,1)]:
    x=torch.randn(M,cin,device='cuda'); wt=ops.pack_conv_weight(torch.randn(cout,cin,kh,kw,device='cuda')*0.05); b=torch.randn(cout,device='cuda')
    for _ in range(3): ops.conv2d(x,wt,b,P,h,w,cout,kh,kw,act="relu")
    torch.cuda.synchronize()
