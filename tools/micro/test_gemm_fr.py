"""The probe kernel of tools/micro/gemm_fr.hip (tile 17) against the product's persistent kernel (tile 16), bit for bit.  Not part of the product test
suite: the probe is linked into diagnostic builds only.
    python tools/build_variant.py fr -DWG_GEMM_FR && WG_LIB=walkgpt_amd/_abl/lib_fr.so python -m pytest tools/micro/test_gemm_fr.py -q"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from walkgpt_amd import ops  # noqa: E402

dev = "cuda:0"


def _ref_act(y, act):
    if act == ops.ACT_GELU:
        return torch.nn.functional.gelu(y)
    if act == ops.ACT_QUICK_GELU:
        return y * torch.sigmoid(1.702 * y)
    return y


@pytest.mark.parametrize("M,N,K", [(1777, 520, 256), (256, 256, 128), (8200, 1024, 1024), (4096, 2304, 768), (3000, 776, 192), (70000, 256, 128)])
@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_GELU, ops.ACT_QUICK_GELU])
def test_gemm_one_barrier_kernel_matches_the_ping_pong_kernel_bit_for_bit(M, N, K, act):
    """tile 17 (tools/micro/gemm_fr.hip, the experimental one-barrier-per-slab persistent kernel: the leading wave group's load half-phase runs beside the
    trailing group's 64-MFMA burst by wave priority) sums in the same order as tile 16: identical bits on ragged M / N, several tiles per workgroup
    and every activation; and both stay within the bf16 bound of fp32 torch.  The kernel is a probe: it is linked into diagnostic builds only
    (tools/build_variant.py <tag> -DWG_GEMM_FR, run with WG_LIB=...); against the product library this test has nothing to compare and skips."""
    from walkgpt_amd import _lib
    if not hasattr(_lib.lib(), "wg_gemm_fr_present"):
        pytest.skip("product library: the one-barrier probe kernel is not linked in")
    g = torch.Generator().manual_seed(M + N + K + act)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    o16 = ops.linear(a, w, b, act=act, tile=16)
    o17 = torch.full_like(o16, float("nan"))
    ops.linear(a, w, b, act=act, out=o17, tile=17)
    assert torch.equal(o16, o17)
    ref = _ref_act(a.float() @ w.float().t() + b.float(), act)
    assert (o17.float() - ref).abs().max().item() <= 0.02 * max(1.0, ref.abs().max().item())
