"""GPU parity of every operator of the hot path against the CPU restatement (oracle/), through the
C ABI.  f32: relative L2 error <= 1e-4; bf16: <= 2e-2 (tests/util.py)."""
import numpy as np
import pytest
import torch

from tests.util import TOL, max_err, rel_err, sub

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import ldm_oracle
    return ldm_oracle


@pytest.fixture(scope="module")
def A():
    import diffusionmodel_amd
    return diffusionmodel_amd


def _say(msg):
    print(msg, flush=True)


# ---- schedule + scheduler step ------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["ddpm", "ddim"])
def test_sched_step(A, O, mode):
    torch.manual_seed(0)
    T = 50
    z = torch.randn(3, 8, 6, 10)
    eps = torch.randn_like(z)
    noise = torch.randn_like(z)
    coef = O.ddpm_coefficients(T) if mode == "ddpm" else O.ddim_coefficients(1000, 50)
    for i in (49, 17, 1, 0):
        if mode == "ddpm":
            ref = O.ddpm_step(z, eps, i, coef, noise)
        else:
            ref = O.ddim_step(z, eps, i, coef)
        got = A.sched_step(z.cuda().clone(), eps.cuda(), coef[i], mode, last=(mode == "ddpm" and i == 0),
                           noise=noise.cuda())
        _say(f"sched {mode} i={i} max_err={max_err(got, ref):.3e}")
        # same f32 operation order on both sides: equal up to the rounding of the division
        assert rel_err(got, ref) < 1e-6


def test_sin_time_addition(A, O):
    t = torch.tensor([0, 1, 49, 999, 500], dtype=torch.int64)
    ref = O.sin_time_addition(t, 256)
    got = A.sin_time_addition(t.cuda(), 256)
    _say(f"sin_time_addition max_err={max_err(got, ref):.3e}")
    assert max_err(got, ref) < 2e-4      # sin/cos of arguments up to 999 in f32
    ref64 = O.sin_time_addition(t, 256, dtype=torch.float64)
    assert max_err(got, ref64) < 2e-4


def test_time_addition(A, O):
    from diffusionmodel_amd.weights import synthetic_from_shapes, time_addition_param_shapes
    w = synthetic_from_shapes(time_addition_param_shapes(256), 3)
    t = O.sin_time_addition(torch.tensor([3, 40, 7]), 256)
    ref = O.time_addition({"m." + k: v for k, v in w.items()}, "m", t)
    got = A.TimeAddition(256, weights=w)(t.cuda())
    _say(f"time_addition rel_err={rel_err(got, ref):.3e}")
    assert rel_err(got, ref) < 1e-5


def test_philox_normal(A, O):
    ref = O.philox_normal_nchw(1234, 5, 7, 2, 64, 4, 4)
    got = A.philox_normal(2, 64, 4, 4, seed=1234, chain0=5, step=7)
    _say(f"philox max_err={max_err(got, ref):.3e} mean={got.mean().item():.3f} std={got.std().item():.3f}")
    assert max_err(got, ref) < 1e-4
    big = A.philox_normal(4, 64, 32, 32, seed=9)
    assert abs(big.mean().item()) < 0.01 and abs(big.std().item() - 1.0) < 0.01


# ---- ResidualBlock (conv3x3 + folded BN + SiLU + time add + residual) ---------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C,B,H,W", [(64, 2, 8, 8), (128, 2, 4, 4), (64, 1, 20, 36), (256, 3, 16, 16), (64, 2, 64, 64),
                                      (64, 17, 64, 64), (64, 5, 48, 160)])
def test_resblock(A, O, dtype, C, B, H, W):
    from diffusionmodel_amd.weights import resblock_param_shapes, synthetic_from_shapes
    w = synthetic_from_shapes(resblock_param_shapes(C, 256), 11)
    torch.manual_seed(1)
    x = torch.randn(B, C, H, W)
    t = torch.randn(B, 256)
    ref = O.residual_block({"m." + k: v for k, v in w.items()}, "m", x, t)
    _say(f"resblock {dtype} C={C} {B}x{H}x{W}")
    got = A.ResiduelBlock(C, 256, dtype=dtype, weights=w)(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("B,H", [(32, 64), (33, 64), (64, 32), (128, 16), (64, 64)])
def test_resblock_fused_kernel(A, O, B, H):
    """64 channels, bf16, 64-pixel-wide images and at least 128 bands of 16 rows: the ResidualBlock runs as ONE kernel
    (conv1 + SiLU + t in LDS, conv2 + residual + SiLU).  Against the CPU restatement, and against the two-launch path,
    which the same module takes for batches below that size (same arithmetic per image: bf16 intermediate, the same
    order of accumulation)."""
    from diffusionmodel_amd import _lib
    from diffusionmodel_amd.weights import resblock_param_shapes, synthetic_from_shapes
    W = 64
    w = synthetic_from_shapes(resblock_param_shapes(64, 256), 13)
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, 64, H, W, generator=g)
    t = torch.randn(B, 256, generator=g)
    _say(f"fused resblock B={B} H={H}")
    m = A.ResiduelBlock(64, 256, dtype="bf16", weights=w)
    got = m(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert "resfuse64_kernel<0>" in _lib.load().ldm_last_kernel().decode()
    n = min(B, 8)                                        # the restatement on a few images, first and last
    idx = list(range(n // 2)) + list(range(B - n // 2, B))
    ref = O.residual_block({"m." + k: v for k, v in w.items()}, "m", x[idx], t[idx])
    e = rel_err(got[idx], ref)
    _say(f"  vs restatement rel_err={e:.3e}")
    assert e < TOL["bf16"]
    step = max(1, 96 // (H // 16))                       # chunks below 128 bands take the two launches
    parts = torch.cat([m(x[i:i + step].cuda().contiguous(), t[i:i + step].cuda().contiguous()) for i in range(0, B, step)])
    torch.cuda.synchronize()
    assert "resfuse" not in _lib.load().ldm_last_kernel().decode()
    e2 = rel_err(got, parts)
    _say(f"  vs two launches rel_err={e2:.3e} max={max_err(got, parts):.3e}")
    assert e2 < 2e-3


# ---- Attention ----------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("C,B,H,W", [(256, 2, 4, 4), (256, 2, 16, 16), (128, 1, 12, 20), (256, 1, 1, 1), (512, 1, 8, 8),
                                      (256, 64, 16, 16), (256, 64, 32, 32), (512, 64, 16, 32), (256, 64, 8, 8), (256, 9, 8, 16)])
def test_attention(A, O, dtype, C, B, H, W):
    # (256, 64, 8, 8), (256, 9, 8, 16): 64 and 128 tokens, bf16: the four-wave streaming kernel with fewer key tiles than
    #   ring stages and a batch that is no multiple of 8 (bf16 at 256 / 512 channels runs with folded projections)
    # (512, 64, 16, 32): 512 tokens x 256 workgroups of 128 queries: the long-sequence kernel at 512 channels (bf16)
    # (256, 64, 16, 16): enough tokens for the weight-stationary 1x1 kernels of the projections (bf16)
    # (256, 64, 32, 32): 1024 tokens x 256 workgroups of 256 queries: the long-sequence attention kernel (bf16)
    from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
    w = synthetic_from_shapes(attention_param_shapes(C), 5)
    torch.manual_seed(2)
    x = torch.randn(B, C, H, W) * 2.0      # larger scores -> a soft-max that is not flat
    ref = O.attention({"m." + k: v for k, v in w.items()}, "m", x)
    _say(f"attention {dtype} C={C} {B}x{H}x{W}")
    got = A.Attention(C, dtype=dtype, weights=w)(x.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    e_branch = rel_err(got - x.cuda(), ref - x)      # without the residual, which dominates the norm
    _say(f"  rel_err={e:.3e} branch={e_branch:.3e}")
    assert e < TOL[dtype]
    assert e_branch < (5e-4 if dtype == "fp32" else 3e-2)


# The token counts the scaled UNet sees at 256x256 latents: 16 384 tokens at 256 channels (its second attention) and
# 4 096 tokens at 512 channels (the innermost one).  bf16 with enough images for 256 workgroups takes the long-sequence
# kernels that run there (8 waves sharing every key/value tile: four-stage ring at 256 channels, two-stage ring at 512);
# one image takes the four-wave kernel; f32 takes the f32 kernel.  The oracle builds the score matrix 2 048 query rows
# at a time (exact soft-max per row).
@pytest.mark.parametrize("dtype,C,B,H,W", [("bf16", 256, 4, 128, 128), ("bf16", 256, 1, 128, 128), ("fp32", 256, 1, 128, 128),
                                            ("bf16", 512, 8, 64, 64), ("bf16", 512, 1, 64, 64), ("fp32", 512, 1, 64, 64)])
def test_attention_at_the_scaled_unet_token_counts(A, O, dtype, C, B, H, W):
    from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
    w = synthetic_from_shapes(attention_param_shapes(C), 5)
    torch.manual_seed(C + B)
    x = torch.randn(B, C, H, W) * 2.0
    _say(f"attention {dtype} C={C} {B}x{H}x{W} ({H * W} tokens)")
    ref = O.attention({"m." + k: v for k, v in w.items()}, "m", x)
    got = A.Attention(C, dtype=dtype, weights=w)(x.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    e_branch = rel_err(got - x.cuda(), ref - x)
    _say(f"  rel_err={e:.3e} branch={e_branch:.3e}")
    assert e < TOL[dtype]
    assert e_branch < (5e-4 if dtype == "fp32" else 3e-2)


# The bf16 block at 256 / 512 channels runs with its projections folded at pack time (keys = the block's input, no output
# projection: q' = x (Wq Wk^T) + Wk bq, v' = x (Wv Wo), constant bv Wo + bo).  LDM_ATTN_FOLD=0 keeps the reference's
# four Linears, a second GPU implementation of the same block: the two must agree with each other and with the oracle.
# Shapes: the three instances of the streaming kernel (four waves x 16 queries incl. a batch that is no multiple of 8 and
# fewer key tiles than ring stages; eight waves x 32 queries; 512 channels).
@pytest.mark.parametrize("C,B,H,W", [(256, 64, 16, 16), (256, 9, 8, 16), (256, 3, 8, 8), (256, 64, 32, 32), (512, 64, 16, 32)])
def test_attention_folded_projections_match_the_unfolded_kernels(A, O, C, B, H, W):
    import os
    from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
    w = synthetic_from_shapes(attention_param_shapes(C), 11)
    torch.manual_seed(7)
    x = torch.randn(B, C, H, W) * 2.0
    ref = O.attention({"m." + k: v for k, v in w.items()}, "m", x)
    mod = A.Attention(C, dtype="bf16", weights=w)
    _say(f"attention folded / unfolded bf16 C={C} {B}x{H}x{W}")
    old = os.environ.get("LDM_ATTN_FOLD")
    try:
        os.environ.pop("LDM_ATTN_FOLD", None)
        folded = mod(x.cuda()).cpu()
        os.environ["LDM_ATTN_FOLD"] = "0"
        plain = mod(x.cuda()).cpu()
    finally:
        if old is None:
            os.environ.pop("LDM_ATTN_FOLD", None)
        else:
            os.environ["LDM_ATTN_FOLD"] = old
    ef, ep = rel_err(folded - x, ref - x), rel_err(plain - x, ref - x)
    ed = rel_err(folded - x, plain - x)
    _say(f"  branch rel_err folded={ef:.3e} unfolded={ep:.3e} folded vs unfolded={ed:.3e}")
    assert not torch.equal(folded, plain)      # the switch did select two implementations
    assert ef < 3e-2 and ep < 3e-2 and ed < 3e-2
    assert rel_err(folded, ref) < TOL["bf16"]


@pytest.mark.parametrize("spike_at", [5, 600, 1023])
def test_attention_running_max_jump_long_sequence(A, O, spike_at):
    """The same property on the long-sequence kernel (bf16, 1024 tokens, 64 images)."""
    from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
    C, B, H, W = 256, 64, 32, 32
    w = synthetic_from_shapes(attention_param_shapes(C), 6)
    torch.manual_seed(40 + spike_at)
    x = torch.randn(B, C, H, W)
    x[:, :, spike_at // W, spike_at % W] *= 12.0
    ref = O.attention({"m." + k: v for k, v in w.items()}, "m", x)
    _say(f"attention spike bf16 long sequence at token {spike_at}")
    got = A.Attention(C, dtype="bf16", weights=w)(x.cuda())
    torch.cuda.synchronize()
    e_branch = rel_err(got - x.cuda(), ref - x)
    _say(f"  branch rel_err={e_branch:.3e}")
    assert e_branch < 3e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("spike_at", [3, 200, 253])
def test_attention_running_max_jump(A, O, dtype, spike_at):
    """One token far larger than the rest: its key dominates every query's soft-max, so the running
    reference of the online soft-max jumps by more than the lazy-rescale threshold at the tile that
    holds the token (first, late and last tile) and the accumulated output must be rescaled there."""
    from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
    C, B, H, W = 256, 2, 16, 16
    w = synthetic_from_shapes(attention_param_shapes(C), 6)
    torch.manual_seed(20 + spike_at)
    x = torch.randn(B, C, H, W)
    x[:, :, spike_at // W, spike_at % W] *= 12.0
    ref = O.attention({"m." + k: v for k, v in w.items()}, "m", x)
    _say(f"attention spike {dtype} at token {spike_at}")
    got = A.Attention(C, dtype=dtype, weights=w)(x.cuda())
    torch.cuda.synchronize()
    e_branch = rel_err(got - x.cuda(), ref - x)
    _say(f"  branch rel_err={e_branch:.3e}")
    assert e_branch < (5e-4 if dtype == "fp32" else 3e-2)


# ---- every convolution kind through the kernel-level operator, at sizes on both sides of each ---------
# ---- specialised kernel's dispatch threshold --------------------------------------------------------
_CONV_CASES = [
    # kind, B, S, cin0, cin1, cout, act, res
    ("conv3x3", 16, 64, 64, 0, 64, None, False),        # weight-stationary 64-channel kernel, plain
    ("conv3x3", 16, 64, 64, 0, 64, "silu", True),       # ... with residual + SiLU
    ("conv3x3", 2, 24, 64, 0, 64, "relu", True),        # generic kernel, ragged tiles
    ("conv3x3", 64, 32, 128, 0, 128, "silu", True),     # generic, 16-row x 128-channel tiles
    ("conv3x3", 64, 16, 256, 0, 256, None, False),      # generic, 16-row x 64-channel tiles
    ("conv3x3", 3, 16, 256, 0, 256, None, True),
    ("conv3x3", 5, 4, 256, 0, 256, None, False),        # images smaller than a tile, narrow channel blocks
    ("convT4x4_s2", 64, 16, 256, 0, 128, None, False),
    ("convT4x4_s2", 4, 10, 128, 0, 64, "relu", False),
    ("conv1x1", 64, 16, 256, 0, 768, None, False),      # weight-stationary 1x1 kernel
    ("conv1x1", 64, 16, 256, 0, 256, None, True),       # ... with residual
    ("conv1x1", 2, 12, 256, 0, 768, None, False),       # generic 1x1
    ("conv1x1", 8, 32, 128, 128, 128, None, False),     # two sources (the concatenation is never built)
    ("conv1x1", 1, 3, 1024, 0, 512, "silu", True),      # a few pixels: the skinny kernel, one 16-pixel fragment, ragged
    ("conv1x1", 3, 5, 128, 128, 192, "relu", False),    # ... 75 pixels over two workgroup rows, two sources
    ("conv1x1", 7, 8, 256, 0, 64, None, True),          # ... 448 pixels
]


@pytest.mark.parametrize("kind,B,S,cin0,cin1,cout,act,res", _CONV_CASES)
def test_conv_ops_against_torch(A, kind, B, S, cin0, cin1, cout, act, res):
    """Each burn layer behind the tap-conv kernels (src/main.rs:348-366, 423-452, 460-483) against torch on
    the same bf16-rounded operands; only the output rounding differs."""
    import torch.nn.functional as F
    from diffusionmodel_amd.ops import ConvOp, from_nhwc, to_nhwc
    g = torch.Generator().manual_seed(B * 1000 + S + cout)
    cin = cin0 + cin1
    taps = {"conv3x3": 9, "convT4x4_s2": 4, "conv1x1": 1}[kind]
    sc = 1.0 / (taps * cin) ** 0.5
    if kind == "convT4x4_s2":
        w = (torch.randn(cin, cout, 4, 4, generator=g) * sc).bfloat16().float()
    elif kind == "conv1x1":
        w = (torch.randn(cout, cin, generator=g) * sc).bfloat16().float()
    else:
        w = (torch.randn(cout, cin, 3, 3, generator=g) * sc).bfloat16().float()
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, S, S, generator=g).bfloat16().float()
    if kind == "convT4x4_s2":
        ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    elif kind == "conv1x1":
        ref = F.conv2d(x, w.reshape(cout, cin, 1, 1), b)
    else:
        ref = F.conv2d(x, w, b, padding=1)
    r = None
    if res:
        r = torch.randn(ref.shape, generator=g).bfloat16().float()
        ref = ref + r
    if act == "silu":
        ref = F.silu(ref)
    elif act == "relu":
        ref = F.relu(ref)
    _say(f"{kind} B={B} S={S} {cin0}+{cin1}->{cout} act={act} res={res}")
    op = ConvOp(kind, w, b, dtype="bf16", cin1=cin1)
    x0 = to_nhwc(x[:, :cin0], "bf16").cuda()
    x1 = to_nhwc(x[:, cin0:], "bf16").cuda() if cin1 else None
    got = from_nhwc(op(x0, x1, res=None if r is None else to_nhwc(r, "bf16").cuda(), act=act))
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < 4e-3


# ---- stride-2 convolutions at the sizes of the weight-stationary kernels --------------------------
# (16, 64, 128, 256): two tiles per output row, so the 128 -> 256 kernel sees tiles on and off the left border
@pytest.mark.parametrize("B,S,cin,cout", [(16, 64, 64, 128), (64, 32, 128, 256), (70, 32, 128, 256), (3, 32, 128, 256), (16, 64, 128, 256)])
def test_conv_stride2_large(A, B, S, cin, cout):
    """conv3x3 / stride 2 (down2, down3; src/main.rs:434-452) through the kernel-level operator at batch
    sizes that reach the weight-stationary kernels (and one that does not), against torch on the same
    bf16-rounded operands."""
    import torch.nn.functional as F
    from diffusionmodel_amd.ops import ConvOp, from_nhwc, to_nhwc
    g = torch.Generator().manual_seed(B + cin)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3.0 * cin ** 0.5))).bfloat16().float()
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, S, S, generator=g).bfloat16().float()
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    _say(f"conv3x3_s2 B={B} S={S} {cin}->{cout}")
    got = from_nhwc(ConvOp("conv3x3_s2", w, b, dtype="bf16")(to_nhwc(x, "bf16").cuda()))
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < 4e-3      # bf16 rounding of the output only


# ---- f32 3x3 convolutions over a few images: the skinny 3x3 kernel --------------------------------
@pytest.mark.parametrize("stride,B,S,cin,cout,act,res", [(1, 1, 16, 256, 256, "silu", True), (1, 2, 22, 64, 64, None, False),
                                                         (1, 3, 7, 128, 128, "relu", True), (2, 1, 32, 128, 256, None, False),
                                                         (2, 2, 14, 64, 128, None, False), (1, 16, 64, 64, 64, None, False)])
def test_conv3x3_f32_small_batches(A, stride, B, S, cin, cout, act, res):
    """f32 conv3x3 (stride 1 and 2, padding 1; src/main.rs:348-366, 434-452) at one to three images, ragged against
    the 4x16-pixel tile, with and without activation and residual, against torch f32; the last case is large enough
    for the tiled kernel (the same operator, the other kernel)."""
    import torch.nn.functional as F
    from diffusionmodel_amd.ops import ConvOp, from_nhwc, to_nhwc
    g = torch.Generator().manual_seed(B * 100 + S + cout + stride)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3.0 * cin ** 0.5))
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, S, S, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    r = None
    if res:
        r = torch.randn(ref.shape, generator=g)
        ref = ref + r.double()
    ref = F.silu(ref) if act == "silu" else F.relu(ref) if act == "relu" else ref
    _say(f"conv3x3 f32 stride={stride} B={B} S={S} {cin}->{cout} act={act} res={res}")
    op = ConvOp("conv3x3" if stride == 1 else "conv3x3_s2", w, b, dtype="fp32")
    got = from_nhwc(op(to_nhwc(x, "fp32").cuda(), None, res=None if r is None else to_nhwc(r, "fp32").cuda(), act=act))
    torch.cuda.synchronize()
    e = rel_err(got, ref.float())
    _say(f"  rel_err={e:.3e}")
    assert e < 1e-5


# ---- forward process and loss value ---------------------------------------------------------------
def test_add_noise_bit_exact(A, O):
    T, B = 1000, 6
    m = A.DiffusionModel(64, 3, T, 256, dtype="fp32", weights=A.make_synthetic_weights(A.LDMConfig(), 0))
    g = torch.Generator().manual_seed(4)
    x, n = torch.randn(B, 64, 4, 4, generator=g), torch.randn(B, 64, 4, 4, generator=g)
    t = torch.tensor([0, 1, 500, 999, 4000, -2])
    _say("add_noise")
    got, n_back = m.add_noise(x.cuda(), t.cuda(), n.cuda())
    ref = O.add_noise(x, t.tolist(), T, n)
    assert torch.equal(got.cpu(), ref) and torch.equal(n_back.cpu(), n)
    noisy, drawn = m.add_noise(x.cuda(), t.cuda())                    # noise drawn on the device
    assert torch.equal(noisy.cpu(), O.add_noise(x, t.tolist(), T, drawn.cpu()))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_diffusion_loss_value(A, O, dtype):
    """DiffusionModel::forward from the latent on (src/main.rs:641-679), timesteps and noise injected."""
    T, B, S = 1000, 4, 8
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    g = torch.Generator().manual_seed(8)
    z, n = torch.randn(B, 64, S, S, generator=g), torch.randn(B, 64, S, S, generator=g)
    t = torch.tensor([3, 250, 600, 999])
    ref = O.diffusion_loss(w, z, t.tolist(), T, n).item()
    _say(f"diffusion loss {dtype}")
    m = A.DiffusionModel(64, 3, T, 256, dtype=dtype, weights=w)
    got = m.forward(z.cuda(), t.cuda(), n.cuda(), train=False)    # the sampler's semantics; training mode: test_gpu_train.py
    assert got.shape == (1,)
    e = abs(got.item() - ref) / abs(ref)
    _say(f"  loss={got.item():.6f} ref={ref:.6f} rel={e:.3e}")
    assert e < (1e-5 if dtype == "fp32" else 2e-2)
    assert m.forward(z.cuda()).item() > 0.0                            # random timesteps and noise


# ---- VAE decoder and the 8-bit image ------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B", [1, 5, 64, 70])
def test_vae_decode(A, O, dtype, B):
    # 70: more rows than one pass of the wide Linear holds (64); 5: a ragged last row group
    w = A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 21)
    torch.manual_seed(B)
    z = torch.randn(B, 64, 4, 4)
    ref = O.vae_decode(w, "", z)
    _say(f"vae decode {dtype} B={B}")
    got = A.Decoder(3, 64, dtype=dtype, weights=w)(z.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e} max_err={max_err(got, ref):.3e}")
    assert got.shape == (B, 3, 64, 64)
    assert e < TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_vae_decode_small_latent(A, O, dtype):
    # latent_dimen 8: K = 128 for the wide Linear (its short weight-fragment batches), one image channel
    w = A.synthetic_from_shapes(A.decoder_param_shapes(1, 8), 3)
    torch.manual_seed(1)
    z = torch.randn(3, 8, 4, 4)
    ref = O.vae_decode(w, "", z)
    _say(f"vae decode latent 8 {dtype}")
    got = A.Decoder(1, 8, dtype=dtype, weights=w)(z.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert got.shape == (3, 1, 64, 64) and e < TOL[dtype]


def test_vae_decode_many_image_channels(A, O):
    # more than 4 image channels: the last layer runs on the tap-conv kernel instead of the direct one
    w = A.synthetic_from_shapes(A.decoder_param_shapes(8, 64), 4)
    torch.manual_seed(2)
    z = torch.randn(2, 64, 4, 4)
    ref = O.vae_decode(w, "", z)
    _say("vae decode 8 image channels bf16")
    got = A.Decoder(8, 64, dtype="bf16", weights=w)(z.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert got.shape == (2, 8, 64, 64) and e < TOL["bf16"]


def test_to_rgb8_bit_exact(A, O):
    torch.manual_seed(5)
    img = torch.tanh(torch.randn(3, 3, 64, 64) * 1.5)
    img[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, 0.0, 0.5, 2.0, -3.0, float("nan"), 0.99999])
    _say("to_rgb8")
    got = A.to_rgb8(img.cuda()).cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == (3, 64, 64, 3)
    assert np.array_equal(got, O.to_rgb8(img.numpy()))
    one = A.to_rgb8(torch.zeros(2, 1, 5, 7).cuda()).cpu().numpy()     # other channel counts and sizes
    assert one.shape == (2, 5, 7, 1) and (one == 127).all()


def test_sample_images_end_to_end(A, O):
    """denoising_process + vae.decode as the reference's denoising_process returns (src/main.rs:681-795),
    f32, 4x4 latents, injected noise: pixels against the restatement."""
    cfg = A.LDMConfig()
    w = A.make_synthetic_weights(cfg, 0)
    wd = A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 21)
    T, B = 10, 2
    g = torch.Generator().manual_seed(77)
    z0, noise = torch.randn((B, 64, 4, 4), generator=g), torch.randn((T, B, 64, 4, 4), generator=g)
    ref = O.vae_decode(wd, "", O.denoising_process(w, z0, T, noise=noise))
    _say("sample_images fp32")
    m = A.DiffusionModel(64, 3, T, 256, dtype="fp32", weights=w, decoder_weights=wd)
    img = m.sample_images(B, z0=z0.cuda(), noise=noise.cuda())
    e = rel_err(img, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < 1e-4
    d = np.abs(A.to_rgb8(img).cpu().numpy().astype(np.int32) - O.to_rgb8(ref.numpy()).astype(np.int32))
    assert d.max() <= 1


# ---- full UNet forward --------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("Cl,B,S", [(64, 4, 4), (64, 2, 16), (8, 2, 8), (64, 1, 64), (64, 16, 64)])
def test_unet_forward(A, O, dtype, Cl, B, S):
    # (64, 16, 64) is large enough for the weight-stationary 64-channel kernels and the fused
    # up-sampling block (bf16); the smaller cases run the generic tap-conv kernel everywhere
    te_all = [7, 31, 0, 49, 3, 12, 25, 44, 1, 2, 8, 16, 33, 40, 48, 5]
    cfg = A.LDMConfig(in_channels=Cl)
    w = A.make_synthetic_weights(cfg, 0)
    torch.manual_seed(3)
    x = torch.randn(B, Cl, S, S)
    te = O.sin_time_addition(torch.tensor(te_all[:B]), 256)
    ref = O.unet_forward(w, x, te)
    _say(f"unet {dtype} Cl={Cl} B={B} S={S}")
    net = A.UNet(Cl, 256, dtype=dtype, weights=w)
    got = net(x.cuda(), te.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_unet_scaled(A, O, dtype):
    cfg = A.LDMConfig.scaled()
    w = A.make_synthetic_weights(cfg, 0)
    torch.manual_seed(4)
    x = torch.randn(1, 64, 16, 16)
    te = O.sin_time_addition(torch.tensor([900]), 256)
    ref = O.unet_forward(w, x, te, scaled_attention=True)
    _say(f"unet scaled {dtype}")
    net = A.UNet(64, 256, channels=cfg.channels, scaled_attention=True, dtype=dtype, weights=w)
    got = net(x.cuda(), te.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("dtype,B,S", [("fp32", 1, 64), ("bf16", 1, 64), ("bf16", 2, 128)])
def test_unet_scaled_larger_latents(A, O, dtype, B, S):
    """The scaled UNet (channels 128/256/512, second attention after up_res1) against the oracle at 64x64 latents
    (1 024 and 256 tokens) and at 128x128 (4 096 and 1 024 tokens)."""
    cfg = A.LDMConfig.scaled()
    w = A.make_synthetic_weights(cfg, 0)
    torch.manual_seed(40 + S)
    x = torch.randn(B, 64, S, S)
    te = O.sin_time_addition(torch.tensor([900, 120][:B]), 256)
    _say(f"unet scaled {dtype} B={B} S={S}")
    ref = O.unet_forward(w, x, te, scaled_attention=True)
    net = A.UNet(64, 256, channels=cfg.channels, scaled_attention=True, dtype=dtype, weights=w)
    got = net(x.cuda(), te.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert e < TOL[dtype]


def test_unet_scaled_at_the_config4_latent_size_against_the_oracle(A, O):
    """BASELINE config 4's network at its own latent size: the scaled UNet on [1,64,256,256], bf16, against the CPU
    restatement (16 384- and 4 096-token attention; the restatement builds its score matrices 2 048 query rows at a
    time)."""
    cfg = A.LDMConfig.scaled()
    w = A.make_synthetic_weights(cfg, 0)
    x = torch.randn(1, 64, 256, 256, generator=torch.Generator().manual_seed(2560))
    te = O.sin_time_addition(torch.tensor([640]), 256)
    _say("unet scaled 256x256 bf16 vs oracle")
    with torch.no_grad():
        ref = O.unet_forward(w, x, te, scaled_attention=True)
    net = A.UNet(64, 256, channels=cfg.channels, scaled_attention=True, dtype="bf16", weights=w)
    got = net(x.cuda(), te.cuda())
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"  rel_err={e:.3e}")
    assert torch.isfinite(got).all() and e < TOL["bf16"]


@pytest.mark.parametrize("B", [4, 16])
def test_unet_scaled_full_size_properties(A, B):
    """Scaled UNet at 256x256 latents, bf16, four images and sixteen (the share of one GPU in the 8-GPU run of config
    4; 16 384- and 4 096-token attention on the long-sequence kernels): finite; the whole batch equals its two halves
    (which take other attention kernels: another summation order, nothing else); a DDIM loop replayed from its graph
    equals the eager loop bit for bit, twice."""
    cfg = A.LDMConfig.scaled()
    w = A.make_synthetic_weights(cfg, 0)
    g = torch.Generator().manual_seed(256)
    x = torch.randn(B, 64, 256, 256, generator=g).cuda()
    te = A.sin_time_addition(torch.tensor(([999, 500, 20, 0] * 4)[:B]).cuda(), 256)
    _say(f"unet scaled 256x256 bf16, {B} images: batch slices")
    net = A.UNet(64, 256, channels=cfg.channels, scaled_attention=True, dtype="bf16", weights=w)
    whole = net(x, te)
    parts = torch.cat([net(x[i:i + B // 2].contiguous(), te[i:i + B // 2].contiguous()) for i in (0, B // 2)])
    torch.cuda.synchronize()
    e = rel_err(whole, parts)
    _say(f"  whole batch vs halves: rel_err={e:.3e}")
    assert torch.isfinite(whole).all()
    assert e < 1e-2
    del net
    _say("unet scaled 256x256 bf16: graph vs eager")
    m = A.DiffusionModel(64, 3, 1000, 256, cfg.channels, True, dtype="bf16", weights=w)
    eager = m.ddim_sample(B, 3, spatial=256, z0=x, use_graph=False)
    torch.cuda.synchronize()
    assert torch.isfinite(eager).all()
    for _ in range(2):
        zg = m.ddim_sample(B, 3, spatial=256, z0=x, use_graph=True)
        torch.cuda.synchronize()
        assert torch.equal(zg, eager)


# ---- full benchmark size: properties that need no CPU reference ------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_unet_full_size_batch_slices_agree(A, dtype):
    """[64,64,64,64] (the benchmark's per-GPU batch): chains are independent, so the noise predicted for the
    whole batch equals the noise predicted for its four slices of 16 (which take other kernels and tilings:
    the summation order inside a dot product may differ, nothing else), and every slice was checked against
    the restatement at this size in test_unet_forward."""
    cfg = A.LDMConfig()
    net = A.UNet(64, 256, dtype=dtype, weights=A.make_synthetic_weights(cfg, 0))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 64, 64, 64, generator=g).cuda()
    t = torch.randint(0, 1000, (64,), generator=g)
    te = A.sin_time_addition(t.cuda(), 256)
    _say(f"unet full size {dtype}")
    whole = net(x, te)
    parts = torch.cat([net(x[i:i + 16].contiguous(), te[i:i + 16].contiguous()) for i in range(0, 64, 16)])
    torch.cuda.synchronize()
    e = rel_err(whole, parts)
    _say(f"  whole batch vs slices: rel_err={e:.3e}")
    assert torch.isfinite(whole).all()
    assert e < (1e-5 if dtype == "fp32" else 1e-2)
    # permuting the chains permutes the result
    perm = torch.randperm(64, generator=g).cuda()
    again = net(x[perm].contiguous(), te[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(again, whole[perm])


# ---- error behaviour: rejected on the host, before any kernel runs ---------------------------------
def test_invalid_arguments_are_rejected(A):
    from diffusionmodel_amd import _lib
    from diffusionmodel_amd.weights import attention_param_shapes, synthetic_from_shapes
    _say("error paths")
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    net = A.UNet(64, 256, dtype="bf16", weights=w)
    te = torch.zeros(1, 256, device="cuda")
    with pytest.raises(_lib.LdmError, match="multiples of 4"):
        net(torch.zeros(1, 64, 6, 8, device="cuda"), te)                   # H not a multiple of 4
    with pytest.raises(ValueError):
        net(torch.zeros(1, 32, 8, 8, device="cuda"), te)                   # wrong channel count
    with pytest.raises(ValueError):
        net(torch.zeros(1, 64, 8, 8), te)                                  # host tensor
    with pytest.raises(ValueError):
        net(torch.zeros(1, 64, 8, 8, device="cuda", dtype=torch.float16), te)
    bad = dict(w)
    del bad["mid_res1.conv2.bias"]
    with pytest.raises(KeyError):
        A.UNet(64, 256, weights=bad)
    bad = dict(w)
    bad["down1.weight"] = torch.zeros(64, 64, 5, 5)
    with pytest.raises(ValueError):
        A.UNet(64, 256, weights=bad)
    with pytest.raises(_lib.LdmError, match="attention supports"):
        A.Attention(192, weights=synthetic_from_shapes(attention_param_shapes(192), 0))(torch.zeros(1, 192, 4, 4, device="cuda"))
    with pytest.raises(ValueError):
        A.Decoder(3, 64, weights=A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 1))(torch.zeros(2, 64, 8, 8, device="cuda"))
    with pytest.raises(_lib.LdmError):
        A.Decoder(3, 12)                                                   # latent_dimen must be a multiple of 8
    m = A.DiffusionModel(64, 3, 50, 256, weights=w)
    with pytest.raises(ValueError):
        m.denoising_process(2, spatial=4, z0=torch.zeros(3, 64, 4, 4, device="cuda"))
    with pytest.raises(_lib.LdmError):
        m.ddim_sample(2, steps=51, spatial=4)                              # more DDIM steps than timesteps
    with pytest.raises(RuntimeError, match="decoder_weights"):
        m.decode(torch.zeros(1, 64, 4, 4, device="cuda"))
    # the module still works after the rejected calls
    assert torch.isfinite(net(torch.zeros(1, 64, 8, 8, device="cuda"), te)).all()
