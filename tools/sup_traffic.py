"""Launch the two many-tile products of the REINFORCE step on their own (for a rocprofv3 --pmc FETCH_SIZE pass per tile-order
variant, tools/ab_supertile_traffic.sh): logits [6400,1000] @ [1000,12000] (128x128 tiles) and dO2 = dlogits [6400,12000] @
embed_word_W^T ([1000,12000] as it lies; nt96x96 tiles), plus the hoisted LSTM2 input product [8000,1500] @ [1500,4000]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import s2vt_amd
from s2vt_amd import ops
torch.manual_seed(0)
A = torch.randn(6400, 1000, device="cuda"); W = torch.randn(1000, 12000, device="cuda") * 0.1; b = torch.zeros(12000, device="cuda")
dl = torch.randn(6400, 12000, device="cuda")
X = torch.randn(8000, 1500, device="cuda"); W2 = torch.randn(1500, 4000, device="cuda") * 0.1
for _ in range(3):
    ops.gemm([ops.operand(A)], W, b, M=6400)
    ops.gemm_nt([ops.operand(dl)], W, None, M=6400)
    ops.gemm([ops.operand(X)], W2, None, M=8000)
torch.cuda.synchronize()
