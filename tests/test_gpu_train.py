"""GPU parity of the TRAINING-mode forward (the forward of DiffusionModel::forward, src/main.rs:635-679, which the
reference only runs on its autodiff backend, :1318): BatchNorm on the statistics of the batch, mid_drop active, one
timestep per image.  Through the C ABI, against the CPU restatement; f32 <= 1e-4, bf16 <= 2e-2 (relative L2)."""
import pytest
import torch

from tests.util import TOL, rel_err, sub

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


# ---- the kernel-level operator: statistics, normalisation, tail --------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 4, 4), (3, 64, 5, 7), (8, 256, 1, 1), (5, 128, 16, 16), (2, 192, 9, 3),
                                   (64, 64, 64, 64), (64, 256, 16, 16), (1, 512, 64, 64)])
def test_batchnorm_act_statistics_and_tail(A, O, dtype, shape):
    from diffusionmodel_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    # a mean several standard deviations away from zero: E[x^2] - mean^2 in f32 would lose the variance
    x = torch.randn(B, C, H, W, generator=g) * (0.5 + torch.rand(1, C, 1, 1, generator=g)) + 6.0 * torch.randn(1, C, 1, 1, generator=g)
    res = torch.randn(B, C, H, W, generator=g)
    tv = torch.randn(B, C, generator=g)
    p = {"bn.gamma": 0.8 + 0.4 * torch.rand(C, generator=g), "bn.beta": 0.1 * torch.randn(C, generator=g)}
    h = ops.to_nhwc(x.cuda(), dtype)
    xr = ops.from_nhwc(h).cpu()                       # what the kernel sees (bf16-rounded in bf16 mode)
    rr = ops.from_nhwc(ops.to_nhwc(res.cuda(), dtype)).cpu()
    _say(f"bn {dtype} {shape}")
    stats = {}
    ref_a = torch.nn.functional.silu(O.batch_norm_train(p, "bn", xr, stats)) + tv.reshape(B, C, 1, 1)
    ref_b = torch.nn.functional.silu(O.batch_norm_train(p, "bn", xr) + rr)
    out, mean, var = ops.batchnorm_act(h, p["bn.gamma"], p["bn.beta"], act="silu", tvec=tv)
    em, ev = rel_err(mean, stats["bn"][0]), rel_err(var, stats["bn"][1])
    _say(f"  mean {em:.2e} var {ev:.2e}")
    assert em < 1e-6 and ev < 1e-5                    # f32 accumulation on both sides; the kernel merges in f64
    tol = TOL[dtype]
    assert rel_err(ops.from_nhwc(out), ref_a) < (1e-5 if dtype == "fp32" else tol)
    out_b, _, _ = ops.batchnorm_act(h, p["bn.gamma"], p["bn.beta"], act="silu", res=ops.to_nhwc(res.cuda(), dtype))
    assert rel_err(ops.from_nhwc(out_b), ref_b) < (1e-5 if dtype == "fp32" else tol)
    # in place, statistics alone, and twice the same bits (fixed summation order)
    _, mean2, var2 = ops.batchnorm_act(h, p["bn.gamma"], p["bn.beta"], stats_only=True)
    assert torch.equal(mean, mean2) and torch.equal(var, var2)
    h2 = h.clone()
    out_c, _, _ = ops.batchnorm_act(h2, p["bn.gamma"], p["bn.beta"], act="silu", tvec=tv, out=h2)
    assert torch.equal(out_c, out)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_dropout_mask_bit_exact(A, O, dtype):
    """The mask of the kernel's dropout equals the restatement's, element for element; survivors are scaled by 2."""
    from diffusionmodel_amd import ops
    B, C, H, W = 3, 64, 6, 5
    x = 1.0 + torch.rand(B, C, H, W, generator=torch.Generator().manual_seed(2))
    h = ops.to_nhwc(x.cuda(), dtype)
    one, zero = torch.ones(C), torch.zeros(C)
    plain, _, _ = ops.batchnorm_act(h, one, zero)
    for seed, p in ((0, 0.5), (0xDEADBEEF12345678, 0.5), (9, 0.25)):
        got, _, _ = ops.batchnorm_act(h, one, zero, dropout_p=p, seed=seed)
        mask = O.dropout_mask_nchw(seed, B, C, H, W, 1.0 - p)
        kept = ops.from_nhwc(got).cpu() != 0
        nz = ops.from_nhwc(plain).cpu() != 0          # an element equal to its channel's mean is 0 kept or not
        assert nz.float().mean().item() > 0.99 and torch.equal(kept, mask.bool() & nz)
        ref = ops.from_nhwc(plain).cpu() * mask * (1.0 / (1.0 - p))
        assert rel_err(ops.from_nhwc(got), ref) < (1e-6 if dtype == "fp32" else 4e-3)


# ---- ResidualBlock and the UNet ----------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,C,S", [(4, 64, 8), (3, 128, 4), (16, 256, 1), (2, 64, 32)])
def test_resblock_train(A, O, dtype, B, C, S):
    w = A.synthetic_from_shapes(A.modules.resblock_param_shapes(C, 256), 5)
    g = torch.Generator().manual_seed(B + C)
    x, t = torch.randn(B, C, S, S, generator=g), torch.randn(B, 256, generator=g)
    p = {"rb." + k: v for k, v in w.items()}
    stats = {}
    ref = O.residual_block(p, "rb", x, t, train=True, stats=stats)
    _say(f"resblock train {dtype} B={B} C={C} S={S}")
    m = A.ResiduelBlock(C, 256, dtype=dtype, weights=w)
    got, mean, var = m.forward(x.cuda(), t.cuda(), train=True, return_stats=True)
    e = rel_err(got, ref)
    _say(f"  rel={e:.3e}")
    assert e < TOL[dtype]
    stol = 1e-4 if dtype == "fp32" else 2e-2
    assert rel_err(mean[:C], stats["rb.bn1"][0]) < stol and rel_err(var[:C], stats["rb.bn1"][1]) < stol
    assert rel_err(var[C:], stats["rb.bn2"][1]) < stol
    # the inference forward of the same module is untouched by the training one
    assert rel_err(m.forward(x.cuda(), t.cuda()), O.residual_block(p, "rb", x, t)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,S", [(8, 4), (4, 8), (2, 16), (3, 32)])
def test_unet_forward_train(A, O, dtype, B, S):
    """Noise prediction of the training step: 4x4 is the reference's own latent size (batch 8: its smallest), per-image
    timesteps, dropout with a seeded mask; the batch statistics of all fourteen BatchNorms are checked too."""
    cfg = A.LDMConfig()
    w = A.make_synthetic_weights(cfg, 0)
    g = torch.Generator().manual_seed(100 + B + S)
    x = torch.randn(B, 64, S, S, generator=g)
    ts = torch.randint(0, 1000, (B,), generator=g)
    te = O.sin_time_addition(ts, 256)
    stats = {}
    with torch.no_grad():
        ref = O.unet_forward(w, x, te, train=True, dropout_seed=77, stats=stats)
    _say(f"unet train {dtype} B={B} S={S}")
    m = A.UNet(64, 256, dtype=dtype, weights=w)
    got, mean, var = m.forward(x.cuda(), te.cuda(), train=True, dropout_seed=77, return_stats=True)
    e = rel_err(got, ref)
    _say(f"  rel={e:.3e}")
    assert e < TOL[dtype]
    names = ["down_res1", "down_res2", "down_res3", "mid_res1", "mid_res2", "up_res1", "up_res2"]
    off = 0
    stol = 1e-4 if dtype == "fp32" else 3e-2
    for n in names:
        C = stats[n + ".bn1"][0].numel()
        for k, bn in enumerate(("bn1", "bn2")):
            sl = slice(off + k * C, off + (k + 1) * C)
            assert rel_err(var[sl], stats[f"{n}.{bn}"][1]) < stol, (n, bn)
            assert (mean[sl].cpu() - stats[f"{n}.{bn}"][0]).abs().max().item() < stol * (1.0 + stats[f"{n}.{bn}"][1].sqrt().max().item()), (n, bn)
        off += 2 * C
    assert off == mean.numel()
    # same seed, same bits; another seed, another mask; and the inference forward still answers as before
    again = m.forward(x.cuda(), te.cuda(), train=True, dropout_seed=77)
    assert torch.equal(again, got)
    assert not torch.equal(m.forward(x.cuda(), te.cuda(), train=True, dropout_seed=78), got)
    with torch.no_grad():
        assert rel_err(m.forward(x.cuda(), te.cuda()), O.unet_forward(w, x, te)) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_diffusion_loss_training_mode(A, O, dtype):
    """DiffusionModel::forward as the reference runs it (src/main.rs:635-679 on the autodiff backend): the loss value
    and the prediction behind it, with injected timesteps, noise and dropout seed."""
    T, B, S = 1000, 8, 4
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    g = torch.Generator().manual_seed(11)
    z, n = torch.randn(B, 64, S, S, generator=g), torch.randn(B, 64, S, S, generator=g)
    t = torch.tensor([0, 3, 250, 600, 999, 17, 512, 998])
    with torch.no_grad():
        ref, ref_pred = O.diffusion_loss(w, z, t.tolist(), T, n, train=True, dropout_seed=3, return_pred=True)
        ref_eval = O.diffusion_loss(w, z, t.tolist(), T, n)
    m = A.DiffusionModel(64, 3, T, 256, dtype=dtype, weights=w)
    _say(f"training loss {dtype}")
    got = m.forward(z.cuda(), t.cuda(), n.cuda(), dropout_seed=3)
    e = abs(got.item() - ref.item()) / abs(ref.item())
    _say(f"  loss={got.item():.6f} ref={ref.item():.6f} rel={e:.3e}")
    assert got.shape == (1,) and e < TOL[dtype]
    z_noisy, _ = m.add_noise(z.cuda(), t.cuda(), n.cuda())
    pred = m.noise_predict(z_noisy, A.sin_time_addition(t.cuda(), 256), train=True, dropout_seed=3)
    assert rel_err(pred, ref_pred) < TOL[dtype]
    ev = m.forward(z.cuda(), t.cuda(), n.cuda(), train=False)
    assert abs(ev.item() - ref_eval.item()) / abs(ref_eval.item()) < TOL[dtype]
    assert abs(ref.item() - ref_eval.item()) / abs(ref_eval.item()) > 1e-3      # the two modes are different functions
    assert m.forward(z.cuda()).item() > 0.0                                      # random timesteps and noise


def test_scaled_unet_training_forward(A, O):
    """The wider UNet with its second attention (channels 128 / 256 / 512), bf16, per-image timesteps."""
    cfg = A.LDMConfig.scaled()
    w = A.make_synthetic_weights(cfg, 1)
    g = torch.Generator().manual_seed(5)
    B, S = 3, 16
    x = torch.randn(B, cfg.in_channels, S, S, generator=g)
    te = O.sin_time_addition(torch.tensor([5, 500, 900]), 256)
    with torch.no_grad():
        ref = O.unet_forward(w, x, te, scaled_attention=True, train=True, dropout_seed=1)
    m = A.UNet(cfg.in_channels, 256, cfg.channels, True, "bf16", w)
    got = m.forward(x.cuda(), te.cuda(), train=True, dropout_seed=1)
    assert rel_err(got, ref) < TOL["bf16"]
