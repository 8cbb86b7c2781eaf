"""One-rank RCCL process group through chainer_maskrcnn.optimizers.init_process_group (high-priority collective streams): does
the installed torch accept the options, does an all-reduce issued from a high-priority stream complete."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd'))
import torch, torch.distributed as dist
from chainer_maskrcnn.optimizers import init_process_group
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
init_process_group('nccl')
t = torch.ones(1 << 20, device='cuda')
with torch.cuda.stream(torch.cuda.Stream(priority=-1)):
    dist.all_reduce(t)
torch.cuda.synchronize()
print('allreduce ok', float(t[0]))
dist.barrier()
dist.destroy_process_group()
print('done')
