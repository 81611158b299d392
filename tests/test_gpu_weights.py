"""dn_bdd_compose / dn_bdd_extract: the reference's block-diagonal relation weights (rgin.py:114-120) laid out densely, and the
gradient back to the blocks, bit-exact against torch.block_diag (pure data movement)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,B,si,so", [(8, 4, 16, 16), (3, 2, 5, 7), (1, 1, 64, 64), (6, 8, 32, 32)])
def test_bdd_dense_matches_block_diag_forward_and_backward(dtype, R, B, si, so):
    from dummynode4graphlearning_amd import ops
    torch.manual_seed(R * 100 + B)
    w = torch.randn(R, B * si * so).to(dtype).to(DEV).requires_grad_(True)
    dense = ops.bdd_dense(w, R, B, si, so)
    ref = torch.stack([torch.block_diag(*w.detach()[r].view(B, si, so)) for r in range(R)])
    assert dense.shape == (R, B * si, B * so) and torch.equal(dense, ref)
    g = torch.randn_like(dense)
    dense.backward(g)
    want = torch.stack([torch.stack([g[r, b * si:(b + 1) * si, b * so:(b + 1) * so] for b in range(B)]) for r in range(R)])
    assert torch.equal(w.grad, want.reshape(R, -1))
