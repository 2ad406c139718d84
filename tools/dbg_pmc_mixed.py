import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_amd import corpus
dv = torch.device('cuda', 0)
x = corpus.generate_device(sys.argv[1], 64 << 20, dv)
torch.cuda.synchronize()
print('ok', sys.argv[1], int(x.sum().item()))
