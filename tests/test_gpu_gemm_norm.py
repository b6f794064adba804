"""GPU parity of the GEMM and LayerNorm kernels (through the C-ABI) against plain fp32 torch on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from walkgpt_amd import ops


def _ref_act(y, act):
    if act == ops.ACT_GELU:
        return torch.nn.functional.gelu(y)
    if act == ops.ACT_QUICK_GELU:
        return y * torch.sigmoid(1.702 * y)
    if act == ops.ACT_RELU:
        return torch.relu(y)
    return y


@pytest.mark.parametrize("tile", [1, 2, 14, 11, 16])
def test_gemm_exact_integer_asymmetric(dev, tile):
    # small integers: every product and partial sum is exact in bf16 x bf16 -> fp32, so the result must be
    # bit-exact; W is asymmetric so a transposed / permuted fragment map cannot pass.
    g = torch.Generator().manual_seed(1)
    M, N, K = 300, 272, 128
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    w[:, 0] += torch.arange(N).float() % 5
    ref = a @ w.t()
    out = ops.linear(a.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16), out_f32=True, tile=tile)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("M,N,K,tile", [(8200, 1024, 1024, 1), (8200, 3072, 1024, 2), (4096, 768, 3072, 2),
                                        (77, 256, 256, 1), (1, 512, 4096, 1), (1000, 2304, 768, 0),
                                        (8200, 3072, 1024, 14), (4096, 768, 3072, 14), (300, 512, 64, 14), (700, 256, 128, 14),
                                        (5000, 1024, 192, 11), (8200, 3072, 1024, 16), (4096, 768, 3072, 16), (300, 512, 64, 16),
                                        (33000, 1280, 256, 16), (66000, 512, 128, 16)])      # (the last two: several tiles per workgroup at 4 and at 2 slabs)
@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_GELU, ops.ACT_QUICK_GELU, ops.ACT_RELU])
def test_gemm_epilogues(dev, M, N, K, tile, act):
    g = torch.Generator().manual_seed(M + N + K + act)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16)
    ref = _ref_act(a.float() @ w.float().t() + b.float(), act) + r.float()
    out = ops.linear(a.to(dev), w.to(dev), b.to(dev), act=act, residual=r.to(dev), tile=tile)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 0.02 * max(1.0, ref.abs().max().item()), err
    # fp32 output path: only the accumulation order differs
    out32 = ops.linear(a.to(dev), w.to(dev), b.to(dev), act=act, residual=r.to(dev), out_f32=True, tile=tile)
    err32 = (out32.cpu() - ref).abs().max().item()
    assert err32 <= 2e-3, err32


def test_gemm_tile_order_for_operands_larger_than_the_infinity_cache(dev):
    """A > 128 MB with weights past 3 MB: the persistent kernel walks XCD-aligned column blocks (here 4 tile columns of 8) instead of the L2-sized ones;
    a tile order is a permutation of the same tiles, so every order must give the same bits as the plain row-major one of a smaller row count."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 66000, 2048, 1024
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    out = ops.linear(a, w, b, tile=16)
    ref = a.float() @ w.float().t() + b.float()
    assert (out.float() - ref).abs().max().item() <= 0.02 * max(1.0, ref.abs().max().item())
    # rows 0 .. 32767 alone are 67 MB: the old rule (blocks of 3) -- same tiles, same arithmetic per tile
    part = ops.linear(a[:32768], w, b, tile=16)
    assert torch.equal(part, out[:32768])


@pytest.mark.parametrize("M,N,K", [(8200, 1024, 1024), (129, 128, 64), (144, 256, 128), (263, 384, 192), (8200, 3072, 1024)])
@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_QUICK_GELU])
def test_gemm_tail_absorbing_tiles(dev, M, N, K, act):
    """tile 12: the last 128-row tile also computes the M % 128 <= 16 leftover rows (CLIP's 8*1025 token rows)."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16)
    ref = _ref_act(a.float() @ w.float().t() + b.float(), act) + r.float()
    out = ops.linear(a.to(dev), w.to(dev), b.to(dev), act=act, residual=r.to(dev), tile=12)
    err = (out.float().cpu() - ref).abs()
    assert err.max().item() <= 0.02 * max(1.0, ref.abs().max().item())
    assert err[-16:].max().item() <= 0.02 * max(1.0, ref.abs().max().item())   # the absorbed rows
    # exact small-integer check of the absorbed rows' fragment mapping
    ai = torch.randint(-3, 4, (M, K), generator=g).float()
    wi = torch.randint(-3, 4, (N, K), generator=g).float()
    wi[:, 0] += torch.arange(N).float() % 5
    oi = ops.linear(ai.to(dev, torch.bfloat16), wi.to(dev, torch.bfloat16), tile=12)
    refi = (ai @ wi.t())
    assert torch.equal(oi.float().cpu(), refi.to(torch.bfloat16).float())


def test_gemm_residual_row_modulo_and_strided(dev):
    g = torch.Generator().manual_seed(5)
    M, N, K = 512, 128, 192
    a_full = torch.randn(M, 3 * K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    pos = torch.randn(128, N, generator=g).to(torch.bfloat16)
    a = a_full[:, K:2 * K]
    ref = a.float() @ w.float().t() + pos.float().repeat(4, 1)
    ad = a_full.to(dev)[:, K:2 * K]
    out = ops.linear(ad, w.to(dev), residual=pos.to(dev), res_row_mod=128, out_f32=True)
    assert (out.cpu() - ref).abs().max().item() <= 2e-3


@pytest.mark.parametrize("M,N,K", [(13, 1, 128), (7, 4, 256), (20, 32, 256), (5, 36, 100)])
def test_gemm_rowwave_fallback(dev, M, N, K):
    g = torch.Generator().manual_seed(9)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = torch.relu(a.float() @ w.float().t() + b.float())
    out = ops.linear(a.to(dev), w.to(dev), b.to(dev), act=ops.ACT_RELU, out_f32=True)
    assert (out.cpu() - ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 4096, 4096), (16, 256, 4096), (3, 256, 256), (14, 1024, 128)])
def test_gemm_skinny_rows(dev, M, N, K):
    """M <= 16 (the [SEG] rows through text_hidden_fcs): one workgroup per 16 output columns, four waves splitting K."""
    assert ops.gemm_tile_for(M, N, K, K, K, N, N) == 5
    g = torch.Generator().manual_seed(11)
    # exact part: small integers, asymmetric W -> bit-exact, and rows >= M of the 16-row MFMA operand must not leak in
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    w[:, 0] += torch.arange(N).float() % 5
    out = ops.linear(a.to(dev, torch.bfloat16), w.to(dev, torch.bfloat16), out_f32=True)
    assert torch.equal(out.cpu(), a @ w.t())
    # epilogue: bias, activation, residual with a row modulo, bf16 and fp32 outputs, a column-slice A operand
    a_full = torch.randn(M, 2 * K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    r = torch.randn(2, N, generator=g).to(torch.bfloat16)
    ref = _ref_act(a_full[:, K:].float() @ w.float().t() + b.float(), ops.ACT_GELU) + r.float().repeat((M + 1) // 2, 1)[:M]
    ad = a_full.to(dev)[:, K:]
    o32 = ops.linear(ad, w.to(dev), b.to(dev), act=ops.ACT_GELU, residual=r.to(dev), res_row_mod=2, out_f32=True)
    o16 = ops.linear(ad, w.to(dev), b.to(dev), act=ops.ACT_GELU, residual=r.to(dev), res_row_mod=2)
    assert (o32.cpu() - ref).abs().max().item() <= 2e-3
    assert (o16.float().cpu() - ref).abs().max().item() <= 3e-2


@pytest.mark.parametrize("M,N,K", [(1, 512, 4096), (8, 512, 4096), (16, 512, 5120), (5, 256, 256), (20, 512, 4096), (112, 512, 4096),
                                   (128, 256, 512), (130, 512, 4096)])
def test_layernorm_linear_skinny(dev, M, N, K):
    """LayerNorm fused in front of the skinny GEMM (head of text_hidden_fcs[0]), up to eight 16-row blocks; M = 130 takes the two-kernel
    route."""
    g = torch.Generator().manual_seed(13)
    x = (torch.randn(M, K, generator=g) * 2.0 + 0.7).to(torch.bfloat16)
    x[:, 3] += 40.0                                         # an outlier channel, as LLM hidden states have
    gam = (1.0 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16)
    bet = (0.1 * torch.randn(K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    ln = torch.nn.functional.layer_norm(x.float(), (K,), gam.float(), bet.float(), 1e-5)
    ref = _ref_act(ln.to(torch.bfloat16).float() @ w.float().t() + b.float(), ops.ACT_GELU)   # (the normalised rows enter the MFMA as bf16)
    out = ops.layernorm_linear(x.to(dev), gam.to(dev), bet.to(dev), 1e-5, w.to(dev), b.to(dev), act=ops.ACT_GELU, out_f32=True)
    assert out.shape == (M, N)
    assert (out.cpu() - ref).abs().max().item() <= 2e-2     # a bf16 ulp of a normalised value (|x^| up to ~20) moving one product
    assert ((out.cpu() - ref).norm() / ref.norm()).item() < 2e-3
    # the same through the weight matrix in fragment order: identical arithmetic, identical bits
    wt = ops.tile_weight(w.to(dev))
    out_t = ops.layernorm_linear(x.to(dev), gam.to(dev), bet.to(dev), 1e-5, w.to(dev), b.to(dev), act=ops.ACT_GELU, out_f32=True, weight_tiled=wt)
    assert torch.equal(out_t, out)
    # and without the LayerNorm
    plain = ops.layernorm_linear(x.to(dev), None, None, 0.0, w.to(dev), b.to(dev), out_f32=True, weight_tiled=wt)
    other = ops.linear(x.to(dev), w.to(dev), b.to(dev), out_f32=True)
    if M <= 16:
        assert torch.equal(plain, other)                    # the same kernel, row-major weights
    else:                                                   # ops.linear takes a tiled kernel there: another summation order
        assert (plain - other).abs().max().item() <= 2e-2 and ((plain - other).norm() / other.norm()).item() < 1e-3


def test_tile_weight_layout(dev):
    """wg_tile_weight_bf16: T[nb][ks][lane][j] = W[16 nb + lane % 16][32 ks + 8 (lane // 16) + j], rows beyond N zero; K-slices tile on their own."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(36, 128, generator=g).to(torch.bfloat16)
    t = ops.tile_weight(w.to(dev)).cpu()
    assert t.shape == (3, 4, 64, 8)
    wp = torch.zeros(48, 128, dtype=torch.bfloat16)
    wp[:36] = w
    ref = wp.reshape(3, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).reshape(3, 4, 64, 8)
    assert torch.equal(t, ref)
    t2 = ops.tile_weight(w.to(dev), 2).cpu()
    assert t2.shape == (2, 3, 2, 64, 8)
    for s in range(2):
        assert torch.equal(t2[s], wp[:, 64 * s:64 * s + 64].reshape(3, 16, 2, 4, 8).permute(0, 2, 3, 1, 4).reshape(3, 2, 64, 8))


@pytest.mark.parametrize("M,D,eps", [(4100, 768, 1e-6), (1025, 1024, 1e-5), (37, 256, 1e-6), (9, 4096, 1e-5),
                                     (3, 5120, 1e-5), (50, 64, 1e-6), (100003, 64, 1e-6), (77, 32, 1e-5), (1001, 128, 1e-5)])
def test_layernorm_rows(dev, M, D, eps):
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(M, D, generator=g) * 3 + 1).to(torch.bfloat16)
    gm = torch.randn(D, generator=g).to(torch.bfloat16)
    bt = torch.randn(D, generator=g).to(torch.bfloat16)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), gm.float(), bt.float(), eps)
    out = ops.layernorm(x.to(dev), gm.to(dev), bt.to(dev), eps)
    # output is rounded to bf16: half an ulp of the largest value
    assert (out.float().cpu() - ref).abs().max().item() <= 0.004 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,N,K", [(8200, 3072, 1024), (8200, 4096, 1024), (32768, 2304, 768), (4096, 3072, 768),
                                   (1025, 3072, 1024), (300, 512, 256), (5125, 1280, 320), (4096, 3840, 1280), (2050, 5120, 1280)])
@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_GELU, ops.ACT_QUICK_GELU])
def test_ln_linear_folded(dev, M, N, K, act):
    """LayerNorm folded into the GEMM (row statistics + persistent kernel epilogue) against fp32 LayerNorm -> linear; rows carry
    a mean several times their spread, so a wrong mean / column-sum term cannot hide.  Small or odd shapes take the two-kernel
    route inside ops.ln_linear and must agree as well."""
    g = torch.Generator().manual_seed(M + N + K + act)
    x = (torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g)) + 3.0 * torch.randn(M, 1, generator=g)).to(torch.bfloat16)
    gamma = (1.0 + 0.3 * torch.randn(K, generator=g)).to(torch.bfloat16)
    beta = (0.2 * torch.randn(K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = _ref_act(torch.nn.functional.linear(torch.nn.functional.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-6),
                                              w.float(), b.float()), act)
    fold = ops.fold_layernorm(gamma.to(dev), beta.to(dev), w.to(dev), b.to(dev))
    out = ops.ln_linear(x.to(dev), fold, 1e-6, act=act)
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= 0.03 * max(1.0, ref.abs().max().item()), err
    st = ops.row_stats(x.to(dev), 1e-6)[:M].cpu()
    xf = x.float()
    assert torch.allclose(st[:, 0], xf.mean(1), atol=1e-5, rtol=1e-5)
    assert torch.allclose(st[:, 1], (xf.var(1, unbiased=False) + 1e-6).rsqrt(), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("M,N,K,Kn", [(6000, 768, 512, 1024), (8200, 1024, 256, 512), (4099, 1280, 320, 1024), (16500, 256, 64, 256)])
def test_row_partials_travel_from_gemm_to_ln_gemm(dev, M, N, K, Kn):
    """wg_gemm_bias_act_stats_bf16 -> wg_gemm_lnp_bias_act_bf16: the GEMM that writes a residual-stream tensor leaves the rows'
    {sum, sum of squares} per 256-column tile, and the LayerNorm-folded GEMM that reads the tensor forms mean / rstd from them
    (image_encoder.py:177-178,191; no statistics pass in between).  Rows carry a mean of several sigma (one-pass variance)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    r = (torch.randn(M, N, generator=g) * (0.5 + torch.rand(M, 1, generator=g)) + 3.0 * torch.randn(M, 1, generator=g)).to(torch.bfloat16)
    xd, wd, bd, rd = x.to(dev), w.to(dev), b.to(dev), r.to(dev)
    y = ops.linear(xd, wd, bd, residual=rd, row_partials=True)
    assert hasattr(y, "_wg_row_partials"), "the shape must take the statistics-producing kernel"
    plain = ops.linear(xd, wd, bd, residual=rd)
    assert torch.equal(y, plain)                                     # same arithmetic, same stores
    part, mpad = y._wg_row_partials[0], y._wg_row_partials[1]
    assert part.shape == (N // 256, mpad, 2) and mpad % 256 == 0 and mpad >= M
    yt = y.float().view(M, N // 256, 256)
    assert torch.allclose(part[:, :M, 0].t(), yt.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[:, :M, 1].t(), (yt * yt).sum(-1), rtol=1e-5, atol=1e-3)
    # consumer: the same LayerNorm-folded GEMM with statistics from the partial sums / from the statistics pass / in fp32 torch
    gamma = (1.0 + 0.3 * torch.randn(N, generator=g)).to(torch.bfloat16)
    beta = (0.2 * torch.randn(N, generator=g)).to(torch.bfloat16)
    w2 = (torch.randn(Kn, N, generator=g) / N ** 0.5).to(torch.bfloat16)
    b2 = torch.randn(Kn, generator=g).to(torch.bfloat16)
    fold = ops.fold_layernorm(gamma.to(dev), beta.to(dev), w2.to(dev), b2.to(dev))
    with ops.time_gemms() as rec:
        got = ops.ln_linear(y, fold, 1e-6, act=ops.ACT_GELU)
    assert [k for k, *_ in rec] == [17]                              # one launch: no row-statistics pass, no two-kernel route
    ref = _ref_act(torch.nn.functional.linear(torch.nn.functional.layer_norm(y.float().cpu(), (N,), gamma.float(), beta.float(), 1e-6),
                                              w2.float(), b2.float()), ops.ACT_GELU)
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= 0.03 * max(1.0, ref.abs().max().item()), err
    via_pass = ops.ln_linear(plain, fold, 1e-6, act=ops.ACT_GELU)      # `plain` carries no partial sums: statistics pass + GEMM
    d = (got.float() - via_pass.float()).abs().max().item()
    assert d <= 0.02 * max(1.0, ref.abs().max().item()), d
    # a view or an in-place edit drops the hand-over instead of using stale sums
    y2 = y.clone()
    y2._wg_row_partials = y._wg_row_partials[:2] + (y2._version,) + y._wg_row_partials[3:]
    y2.mul_(2.0)
    assert torch.equal(ops.ln_linear(y2, fold, 1e-6), ops.ln_linear(y2.clone(), fold, 1e-6))


def test_row_partials_are_refused_for_shapes_the_kernel_cannot_take(dev):
    L = ops._lib.lib()
    assert L.wg_gemm_row_partials_supported(32768, 768, 768, 768, 768, 768) == 1
    assert L.wg_gemm_row_partials_supported(32768, 2304, 768, 768, 768, 2304) == 0     # wider than five 256-column tiles
    assert L.wg_gemm_row_partials_supported(32768, 320, 768, 768, 768, 320) == 0       # not whole column tiles
    x = torch.randn(64, 64, device=dev).to(torch.bfloat16)
    w = torch.randn(256, 64, device=dev).to(torch.bfloat16)
    y = ops.linear(x, w, row_partials=True)                                            # too few rows for the persistent kernel: plain GEMM
    assert not hasattr(y, "_wg_row_partials")
    part = torch.empty(1, 256, 2, device=dev)
    rc = L.wg_gemm_bias_act_stats_bf16(x.data_ptr(), 64, w.data_ptr(), 64, None, None, 0, 0, y.data_ptr(), 256, 64, 256, 64, 0, part.data_ptr(), 256, None)
    assert rc != 0 and b"persistent" in L.wg_last_error()


def test_epilogue_gelu_through_an_identity_gemm(dev):
    """The bf16 epilogue's GELU alone: x @ I with the activation fused, on the persistent 256 x 256 kernel (plain and LayerNorm-free paths share
    wg_act2e), against torch's erf GELU of the same bf16 inputs.  The fitted form is within 2.6e-5 of the erf form before the rounding to bf16, so the
    stored value is within that + half a bf16 step of the exact one; |x| up to 12 covers both saturated ends and the clamp of x^2."""
    M, K = 2048, 256
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, K, generator=g) * 3.0).to(torch.bfloat16)
    x[0, :] = torch.linspace(-12, 12, K).to(torch.bfloat16)
    x[1, :] = torch.linspace(-0.01, 0.01, K).to(torch.bfloat16)
    eye = torch.eye(K).to(torch.bfloat16)
    y = ops.linear(x.to(dev), eye.to(dev), act=ops.ACT_GELU, tile=16).float().cpu()
    ref = torch.nn.functional.gelu(x.double())
    half_step = ref.abs().clamp_min(2.0 ** -126).log2().floor().exp2() * 2.0 ** -8      # half a bf16 step at the reference's magnitude
    assert bool(((y.double() - ref).abs() <= 2.6e-5 + half_step * 1.0001).all())
    assert torch.isfinite(y).all() and abs(float(y[0, 0])) < 1e-12 and float(y[0, -1]) == 12.0

