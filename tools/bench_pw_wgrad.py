"""Streaming 1x1 backward-weight kernel (ur_conv1x1_pw_wgrad) against the generic split-K GEMM (ur_conv2d_wgrad) on the
network's skinny full-resolution layers; the last tuple element is the channel pitch of x (a window of a wider tensor).
URESNET_PW_WGRAD_WGS=<n> sets the number of workgroups (= slabs) of the streaming kernel."""
import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo")
import uresnet_amd
from uresnet_amd import _lib as lib
dev="cuda:0"
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (n,h,w,cin,cout,xld) in [(6,512,1024,80,24,80),(6,512,1024,32,16,80),(6,256,512,64,32,64)]:
    x=torch.randn(n,h,w,xld,device=dev).to(torch.bfloat16); dy=torch.randn(n,h,w,cout,device=dev).to(torch.bfloat16)
    d=lib.ConvDesc(lib.UR_BF16,n,h,w,cin,h,w,cout,1,1,1,0,xld,cout)
    dw=torch.empty(cout,cin,device=dev)
    res=[]
    for name in ("ur_conv1x1_pw_wgrad","ur_conv2d_wgrad"):
        ws=torch.empty(lib.query(name+"_workspace",C.byref(d)),dtype=torch.uint8,device=dev)
        f=lambda: lib.call(name,x.data_ptr(),dy.data_ptr(),dw.data_ptr(),ws.data_ptr(),ws.numel(),C.byref(d),st)
        for _ in range(3): f()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1)/20*1e3)
    byt=(n*h*w*(cin+cout)*2)
    print((n,h,w,cin,cout,xld),"pw %.1f us (%.2f TB/s)  generic %.1f us"%(res[0],byt/res[0]/1e6,res[1]),flush=True)
