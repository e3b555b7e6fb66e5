import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29533')
torch.cuda.set_device(0)
dev=torch.device('cuda',0)
torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from inclusivegan_amd.dnnlib.tflib.graphs import GraphedStep
from inclusivegan_amd.dnnlib.tflib.optimizer import allreduce_mean_
from inclusivegan_amd import hip_ops
x=torch.randn(6,128,32,32,device=dev).contiguous(memory_format=torch.channels_last); w=torch.randn(3,3,128,128,device=dev)/34
buf=torch.zeros(1000,device=dev)
def fn():
    y=hip_ops.conv2d_raw(x,w,hip_ops.ConvGeom(3,3,1,1,1,1),(32,32),128)
    buf.add_(y.flatten()[:1000]); return y
st=GraphedStep(fn,True,eager_calls=1,name='t')
for i in range(5):
    st(); t=torch.ones(10,device=dev); torch.distributed.all_reduce(t); allreduce_mean_(buf)
torch.cuda.synchronize(); print('ok', float(buf.sum()), st.graph is not None)
best=torch.full((4,),-1,dtype=torch.int64,device=dev); torch.distributed.all_reduce(best, op=torch.distributed.ReduceOp.MIN); print(best.tolist())
torch.distributed.barrier(); torch.distributed.destroy_process_group()
