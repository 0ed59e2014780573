"""s2vt_gemm_nt_splitk: the order-free data-gradient product C = A @ Wt^T cut into K slabs (train.hip) against float64, for the
library's own slab choice and forced ones, every tile, ragged shapes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(896, 1000, 12000), (1216, 1500, 4000), (320, 500, 4000), (70, 36, 1028), (384, 1000, 260)])
def test_splitk_matches_float64(gpu, M, N, K):
    import torch
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) * 0.1).cuda()
    ref = (A.double() @ Wt.double().t()).cpu().numpy()
    scale = np.abs(ref).max()
    for splits, cfg in [(0, -1), (1, -1), (2, 0), (3, 2), (6, 1), (12, 0), (5, 7)]:
        if splits > 1 and K // splits < 32:
            continue
        out = gpu.gemm_nt_splitk(A, Wt, splits, cfg).cpu().numpy()
        assert np.abs(out - ref).max() <= 2e-5 * scale, (splits, cfg)


def test_splitk_argument_checks(gpu):
    import torch
    import s2vt_amd
    A = torch.randn(64, 512).cuda(); Wt = torch.randn(32, 512).cuda()
    small = torch.empty(64 * 32, device="cuda")                  # room for one slab only
    with pytest.raises(s2vt_amd.S2VTLibraryError):
        gpu.gemm_nt_splitk(A, Wt, 4, -1, slabs=small)
    out = torch.empty(64, 40, device="cuda")[:, :32]            # ldc != N: slabs cannot be summed into it
    with pytest.raises(s2vt_amd.S2VTLibraryError):
        gpu.gemm_nt_splitk(A, Wt, 4, -1, out=out)
