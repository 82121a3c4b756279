import sys; sys.path.insert(0,'/root/repo')
import torch
from frad_python_amd import core
lib=core._lib.load()
dev=torch.device('cuda:0')
n=460_783_616
a=torch.empty(n,dtype=torch.uint8,device=dev); b=torch.empty(n,dtype=torch.uint8,device=dev)
s=int(torch.cuda.current_stream().cuda_stream)
for _ in range(20): lib.bench_copy(a.data_ptr(),b.data_ptr(),n,s)
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): lib.bench_copy(a.data_ptr(),b.data_ptr(),n,s)
e1.record(); torch.cuda.synchronize()
print("frad_bench_copy: %.2f TB/s" % (2*n*50/(e0.elapsed_time(e1)*1e-3)/1e12))
for _ in range(20): b.copy_(a)
e0.record()
for _ in range(50): b.copy_(a)
e1.record(); torch.cuda.synchronize()
print("torch copy_: %.2f TB/s" % (2*n*50/(e0.elapsed_time(e1)*1e-3)/1e12))
