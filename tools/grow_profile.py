import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from plp import plp, synth
import torch, numpy as np
fr = synth.replay(1234, 64)
d = torch.from_numpy(fr).cuda().repeat(16,1,1).contiguous()
B=len(d); cap=512
lt = plp.LineFeatureTracker()
kl=torch.zeros((B,cap,68),dtype=torch.uint8,device='cuda'); lb=torch.zeros((B,cap,32),dtype=torch.uint8,device='cuda'); fn=torch.zeros((B,cap,3),dtype=torch.float64,device='cuda'); cn=torch.zeros(B,dtype=torch.int32,device='cuda')
for _ in range(2): lt.extract_batch(d,kl,lb,fn,cn)
torch.cuda.synchronize()
print(B, lt.grow_profile())
