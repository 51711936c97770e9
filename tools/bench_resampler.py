import sys, time
sys.path[:0]=['/root/repo/audio-forge_amd']
import torch, numpy as np
from mic_eq_mi import mic_eq_core as core
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
n=441000
x=torch.randn(B,n,dtype=torch.float64,device='cuda')*0.1
r=core.Resampler(44100,48000)
n_out,blocks=r.plan(n)
y=torch.empty(B,n_out,dtype=torch.float64,device='cuda')
for i in range(3):
    r.process_device(x.data_ptr(),y.data_ptr(),n,B,n,n_out,torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); ms=r.last_kernel_ms()
    print(f"B={B} n_out={n_out} {ms:.2f} ms -> {B*n_out/ms*1e3/1e9:.2f} G out frames/s, {B*n_out/ms*1e3/48000:.0f}x RT, f64 FMA rate {B*n_out*512*2/ms*1e3/1e12:.2f} TFLOP/s, audio {B*(n+n_out)*8/ms*1e3/1e9:.1f} GB/s")
