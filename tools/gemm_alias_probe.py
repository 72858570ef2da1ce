import os, sys, torch
sys.path.insert(0, '/root/repo')
from video_rep_learning_amd import _lib
dev='cuda'; M=256*197
st=torch.cuda.current_stream().cuda_stream
for (n,k,epi,name) in ((2304,768,0,'qkv'),(3072,768,1,'fc1'),(768,3072,2,'fc2')):
    A=torch.randn(M,k,device=dev).to(torch.bfloat16); W=(torch.randn(n,k,device=dev)*0.02).to(torch.bfloat16)
    b=torch.randn(n,device=dev); C=torch.empty(M,n,device=dev,dtype=torch.bfloat16); R=torch.zeros(M,n,device=dev)
    for (lda,ldw,tag) in ((k,k,'normal'),(0,k,'A rows aliased'),(k,0,'W rows aliased'),(0,0,'both aliased (all L1/L2 hits)')):
        fn=lambda: _lib.call('mvf_gemm_tc', _lib.BF16, epi, A.data_ptr(), lda, W.data_ptr(), ldw, b.data_ptr(), C.data_ptr(), n, R.data_ptr(), n, None, 0, None, None, 197, M, n, k, st)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/20*1e-3
        print('%-4s %-32s %8.1f us %7.1f TFLOP/s-equivalent'%(name,tag,t*1e6,2.0*M*n*k/t/1e12),flush=True)
