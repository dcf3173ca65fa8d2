"""GPU parity of the sampling loop (denoising_process / DDIM) against the CPU restatement."""
import pytest
import torch

from tests.util import TOL, psnr, rel_err

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


def _model(A, dtype, num_time, weights, cfg=None):
    cfg = cfg or A.LDMConfig()
    return A.DiffusionModel(cfg.in_channels, 3, num_time, cfg.time_embed_dim, cfg.channels, cfg.scaled_attention,
                            dtype=dtype, weights=weights)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_ddpm_chain_with_injected_noise(A, O, dtype):
    """50-step DDPM at the reference's literal latent [B,64,4,4] with identical z0 and per-step noise:
    free-running final latent, and per-step epsilon with teacher forcing (the oracle's z_t fed to both)."""
    T, B, S = 50, 4, 4
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    g = torch.Generator().manual_seed(1234)
    z0 = torch.randn(B, 64, S, S, generator=g)
    noise = torch.randn(T, B, 64, S, S, generator=torch.Generator().manual_seed(4321))
    ref_z, ref_zs, ref_eps = O.denoising_process(w, z0, T, noise=noise, trace=True)
    m = _model(A, dtype, T, w)
    got_z, zt, et = m.denoising_process(B, spatial=S, z0=z0.cuda(), noise=noise.cuda(), use_graph=False,
                                        return_trace=True)
    torch.cuda.synchronize()
    assert rel_err(zt[0], ref_zs[0]) < 1e-6
    e0 = rel_err(et[0], ref_eps[0])
    efin = rel_err(got_z, ref_z)
    _say(f"ddpm {dtype}: eps[0] rel={e0:.3e} final z rel={efin:.3e} psnr={psnr(got_z, ref_z):.1f} dB")
    assert e0 < TOL[dtype]
    # bounds about ten times what is measured on MI355X (f32: 2.1e-7, 150.7 dB; bf16: 1.1e-3, 76.5 dB)
    assert efin < (3e-6 if dtype == "fp32" else 1e-2)
    assert psnr(got_z, ref_z) > (130.0 if dtype == "fp32" else 60.0)
    # teacher forcing: per-step epsilon on the oracle's own z_t
    worst = 0.0
    for k in range(T):
        t = T - 1 - k
        temb = O.sin_time_addition(torch.full((B,), t), 256).cuda()
        e = rel_err(m.noise_predict(ref_zs[k].cuda(), temb), ref_eps[k])
        worst = max(worst, e)
    _say(f"ddpm {dtype}: worst teacher-forced eps rel={worst:.3e}")
    assert worst < (1e-5 if dtype == "fp32" else TOL[dtype])      # measured 9.0e-7 / 5.5e-3
    # graph replay gives the same chain as eager launches, twice
    for _ in range(2):
        zg = m.denoising_process(B, spatial=S, z0=z0.cuda(), noise=noise.cuda(), use_graph=True)
        torch.cuda.synchronize()
        assert torch.equal(zg, got_z)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_ddim_chain(A, O, dtype):
    T, steps, B, S = 1000, 50, 2, 8
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(7))
    ref_z, ref_zs, ref_eps = O.denoising_process(w, z0, T, mode="ddim", ddim_steps=steps, trace=True)
    m = _model(A, dtype, T, w)
    got_z, zt, et = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=False, return_trace=True)
    torch.cuda.synchronize()
    e0 = rel_err(et[0], ref_eps[0])
    efin = rel_err(got_z, ref_z)
    _say(f"ddim {dtype}: eps[0] rel={e0:.3e} final rel={efin:.3e} psnr={psnr(got_z, ref_z):.1f} dB")
    assert e0 < TOL[dtype]
    # measured on MI355X: f32 4.2e-7 / 145.9 dB, bf16 4.8e-3 / 64.7 dB; the bounds are about ten times that
    assert efin < (5e-6 if dtype == "fp32" else 4e-2)
    assert psnr(got_z, ref_z) > (125.0 if dtype == "fp32" else 50.0)
    zg = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=True)
    zg2 = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(zg, got_z) and torch.equal(zg2, got_z)


@pytest.mark.parametrize("mode,dtype", [("ddim", "bf16"), ("ddpm", "fp32")])
def test_benchmark_shape_chain_teacher_forced(A, O, mode, dtype):
    """Per-step epsilon on identical latents and timesteps at the benchmark's latent shape [4,64,64,64]: the oracle
    runs the whole 50-step chain (DDIM over 1000 timesteps in bf16 mode, DDPM over 50 with injected noise in f32
    mode), the GPU predicts the noise for every one of the oracle's z_t (teacher forcing), and a free-running GPU
    chain from the same start gives the final-sample PSNR.

    With synthetic (untrained) weights the network is no denoiser and the DDIM chain, which like the reference never
    clips x0, grows from rms 1 to rms 4e3.  On those latents the f32 oracle itself moves by 1.4e-2 .. 2.3e-2 when only
    its operands (weights and z_t) are rounded to bf16, so the 2e-2 bar cannot hold there for any bf16 implementation.
    The bar is therefore applied where that floor is below 1e-2 (the first ten steps here); elsewhere the GPU may be
    at most twice the floor (measured: at most 1.8 times)."""
    B, S, steps = 4, 64, 50
    T = 1000 if mode == "ddim" else 50
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(1234))
    noise = torch.randn(T, B, 64, S, S, generator=torch.Generator().manual_seed(4321)) if mode == "ddpm" else None
    _say(f"benchmark-shape {mode} {dtype}: oracle chain")
    ref_z, ref_zs, ref_eps = O.denoising_process(w, z0, T, noise=noise, mode=mode, ddim_steps=steps, trace=True)
    ts = O.ddim_timesteps(T, steps) if mode == "ddim" else list(range(T - 1, -1, -1))
    m = _model(A, dtype, T, w)
    w_rounded = {k: v.bfloat16().float() for k, v in w.items()} if dtype == "bf16" else None
    worst, worst_k, worst_ratio, strict = 0.0, -1, 0.0, 0
    for k, t in enumerate(ts):
        temb = O.sin_time_addition(torch.full((B,), t), 256)
        e = rel_err(m.noise_predict(ref_zs[k].cuda(), temb.cuda()), ref_eps[k])
        if e > worst:
            worst, worst_k = e, k
        if dtype == "bf16":
            floor = rel_err(O.unet_forward(w_rounded, ref_zs[k].bfloat16().float(), temb), ref_eps[k])
            if floor < 1e-2:
                strict += 1
                assert e < TOL[dtype], (k, t, e, floor)
            else:
                worst_ratio = max(worst_ratio, e / floor)
                assert e < 2.0 * floor, (k, t, e, floor)
        else:
            assert e < TOL[dtype], (k, t, e)
    _say(f"benchmark-shape {mode} {dtype}: worst teacher-forced eps rel={worst:.3e} at step {worst_k}"
         + (f"; {strict} steps under the 2e-2 bar, elsewhere at most {worst_ratio:.2f} x the operand-rounding floor"
            if dtype == "bf16" else ""))
    assert dtype != "bf16" or strict >= 8
    if mode == "ddim":
        got = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=True)
    else:
        got = m.denoising_process(B, spatial=S, z0=z0.cuda(), noise=noise.cuda(), use_graph=True)
    torch.cuda.synchronize()
    efin, p = rel_err(got, ref_z), psnr(got, ref_z)
    _say(f"benchmark-shape {mode} {dtype}: free-running final rel={efin:.3e} psnr={p:.1f} dB")
    assert torch.isfinite(got).all()
    # measured: f32 2.4e-7 / 153 dB, bf16 1.05e-2 / 57.5 dB
    assert efin < (3e-6 if dtype == "fp32" else 0.1) and p > (130.0 if dtype == "fp32" else 40.0)


def test_benchmark_shape_epsilon_on_forward_process_latents(A, O):
    """The strict bf16 bar at the benchmark's latent shape on latents that look like what a trained model sees:
    z_t = sqrt(abar_t) x0 + sqrt(1 - abar_t) n (unit variance at every timestep), all 50 DDIM timesteps, the GPU and the
    oracle on identical z_t and t."""
    B, S, steps, T = 4, 64, 50, 1000
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    g = torch.Generator().manual_seed(99)
    x0 = torch.randn(B, 64, S, S, generator=g)
    abar = O.get_alphas(T)
    m = _model(A, "bf16", T, w)
    worst = 0.0
    for t in O.ddim_timesteps(T, steps):
        n = torch.randn(B, 64, S, S, generator=g)
        z = x0 * float(abar[t]) ** 0.5 + n * float(1.0 - abar[t]) ** 0.5
        temb = O.sin_time_addition(torch.full((B,), t), 256)
        e = rel_err(m.noise_predict(z.cuda(), temb.cuda()), O.unet_forward(w, z, temb))
        worst = max(worst, e)
        assert e < TOL["bf16"], (t, e)
    _say(f"benchmark-shape forward-process latents bf16: worst eps rel={worst:.3e} over {steps} timesteps")


def test_builtin_generator_chain_matches_oracle_and_is_shard_invariant(A, O):
    """No z0 / noise given: the in-kernel counter-based generator.  The oracle draws the same field
    (Philox + Box-Muller restated in numpy); splitting the batch over 'ranks' changes nothing."""
    T, B, S, seed = 20, 4, 4, 99
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = O.philox_normal_nchw(seed, 0, O.INIT_STEP, B, 64, S, S)
    noise = torch.stack([O.philox_normal_nchw(seed, 0, k, B, 64, S, S) for k in range(T)])
    ref = O.denoising_process(w, z0, T, noise=noise)
    m = _model(A, "fp32", T, w)
    got = m.denoising_process(B, spatial=S, seed=seed, use_graph=True)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"philox chain rel={e:.3e}")
    assert e < 3e-6                 # measured 1.6e-7
    # two "ranks" of 2 chains each, different seed afterwards must change the result (graph reuse)
    parts = [m.denoising_process(2, spatial=S, seed=seed, chain0=c0, use_graph=True) for c0 in (0, 2)]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts), got)
    other = m.denoising_process(B, spatial=S, seed=seed + 1, use_graph=True)
    torch.cuda.synchronize()
    assert not torch.equal(other, got)


def test_long_chain_fp32_drift(A, O):
    """1000-step DDPM in f32 (early steps divide by sqrt(abar) ~ 6e-3 with no clipping)."""
    T, B, S = 1000, 2, 4
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(5))
    noise = torch.randn(T, B, 64, S, S, generator=torch.Generator().manual_seed(6))
    ref = O.denoising_process(w, z0, T, noise=noise)
    m = _model(A, "fp32", T, w)
    got = m.denoising_process(B, spatial=S, z0=z0.cuda(), noise=noise.cuda(), use_graph=True)
    torch.cuda.synchronize()
    e = rel_err(got, ref)
    _say(f"1000-step ddpm fp32 rel={e:.3e} |z|max={ref.abs().max().item():.3e}")
    assert torch.isfinite(got).all()
    assert e < 5e-6                 # measured 4.4e-7


def test_config5_long_chain_at_the_benchmark_latent_size():
    """BASELINE config 5 on the kernels it is benchmarked on: 1000-step f32 DDPM at [4,64,64,64] through the f32 TILED
    convolution kernels (at this batch the library would pick its few-workgroup kernel for the 3x3 layers, so the
    child process sets LDM_SKINNY3_MAXWG=0, which the library reads once per process) and sched_step_kernel with the
    noise read from memory, against the CPU restatement on the same noise."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LDM_SKINNY3_MAXWG="0")
    _say("config 5 long chain (child process)")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "longchain_child.py"), "1000", "4", "64"], env=env,
                       capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    _say(f"  {r}")
    assert r["finite"] and "tapconv_kernel<float" in r["last_conv_kernel"]
    # The parity bar on the restatement's own latents of a dozen steps across the chain: the GPU's epsilon within 1e-4
    # of the f64 restatement (the f32 restatement's own distance from it is reported beside it).  One kind of step is
    # judged differently.  While the untrained network drives the latent from |z| ~ 1 to ~ 1e7 its attention soft-max
    # saturates, and around |z| ~ 1e3 single queries sit on a near-tie between two keys: whichever f32 summation
    # order is used, such a query may pick the other key, which changes that image's prediction by ~1e-3 and nothing
    # else (measured: at step 400, 100 % of the squared error in one image, 96 % in 1 % of the pixels, with the tiled
    # kernels; the few-workgroup kernels pass that step at 2e-6 and the f32 restatement sits at 1e-5).  A step whose
    # error is that localised is accepted up to 2e-3; at most two such steps, all others at the plain bar.
    assert len(r["steps"]) >= 10 and r["sched_step_err"] < 1e-6
    plain = [st for st in r["steps"] if st["gpu_vs_f64"] < 1e-4]
    for st in r["steps"]:
        assert st in plain or (st["err_share_worst_pixels"] > 0.5 and st["gpu_vs_f64"] < 2e-3), st
    assert len(plain) >= len(r["steps"]) - 2
    # free-running end of the chain.  With untrained weights the latent grows to |z| ~ 1e7 and the chain amplifies
    # rounding differences (and every near-tie above): measured 1.5e-4 with the tiled kernels and 1.6e-4 with the
    # few-workgroup kernels, two f32 evaluations that differ only in summation order
    assert r["rel_err"] < 2e-3


def test_ddim_fused_update_large_latent(A, O):
    """bf16 latents large enough for the 64-channel weight-stationary kernel (>= 512 tiles of 8x16): the
    DDIM update and the step advance then run in the epilogue of out_conv.  The traced run (eps_trace
    requested) takes the unfused path; both must agree with each other and with the CPU restatement."""
    T, steps, B, S = 1000, 4, 16, 64
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(11))
    ref_z = O.denoising_process(w, z0, T, mode="ddim", ddim_steps=steps)
    m = _model(A, "bf16", T, w)
    unfused, zt, et = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=False, return_trace=True)
    fused = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=False)
    torch.cuda.synchronize()
    e_pair, e_ref, e_ref_u = rel_err(fused, unfused), rel_err(fused, ref_z), rel_err(unfused, ref_z)
    _say(f"ddim fused: vs unfused rel={e_pair:.3e}  vs oracle fused={e_ref:.3e} unfused={e_ref_u:.3e}")
    # Two bf16 chains differ from each other about as much as each differs from the f32 oracle (measured:
    # 1.1e-2 / 1.2e-2 / 1.2e-2): the first DDIM step divides by sqrt(abar_750) and every later UNet call
    # re-rounds the state to bf16, so different f32 summation orders decorrelate the roundings.
    assert e_pair < 3e-2 and e_ref < 3e-2 and e_ref_u < 3e-2
    for _ in range(2):            # graph replays: the same chain, the step counter restarts every call
        zg = m.ddim_sample(B, steps, spatial=S, z0=z0.cuda(), use_graph=True)
        torch.cuda.synchronize()
        assert torch.equal(zg, fused)


import contextlib
import os


@contextlib.contextmanager
def _tap_kernels_at_4x4():
    """The library reads LDM_DENSE4 at every call: "0" keeps the tap kernels at 4x4 latents."""
    old = os.environ.get("LDM_DENSE4")
    os.environ["LDM_DENSE4"] = "0"
    try:
        yield
    finally:
        if old is None:
            del os.environ["LDM_DENSE4"]
        else:
            os.environ["LDM_DENSE4"] = old


def test_full_size_chain_is_deterministic(A):
    """[64,64,64,64] bf16 DDIM, the benchmark's shape: every launch of the loop runs the weight-stationary and
    fused kernels whose waits are counted by hand.  Replays of the graph, an eager run, and a second model built
    from the same weights must agree bit for bit (a wait that is one short shows up here as a flicker)."""
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(64, 64, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
    m = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
    _say("full-size determinism")
    ref = m.ddim_sample(64, 10, spatial=64, z0=z0, use_graph=True)
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all()
    for _ in range(3):
        again = m.ddim_sample(64, 10, spatial=64, z0=z0, use_graph=True)
        torch.cuda.synchronize()
        assert torch.equal(again, ref)
    eager = m.ddim_sample(64, 10, spatial=64, z0=z0, use_graph=False)
    m2 = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
    other = m2.ddim_sample(64, 10, spatial=64, z0=z0, use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(eager, ref) and torch.equal(other, ref)


@pytest.mark.parametrize("B,S,steps", [(32, 64, 1), (32, 64, 2), (32, 64, 5), (64, 64, 4), (16, 128, 3)])
def test_chained_ddim_steps_equal_the_single_step_kernels(A, B, S, steps):
    """bf16 DDIM on 64-pixel-wide latents with at least 128 bands of 16 rows: out_conv, the DDIM update and the NEXT
    step's down1 run as one kernel (the bf16 copy of the state stays in LDS, the f32 state alternates between two
    buffers).  A sampler created under LDM_DDIMCHAIN=0 runs the same steps as out_conv + update and down1: the same
    arithmetic in the same order, so the final latents must be equal bit for bit - with one step (no chained launch at
    all), two (one chained launch, the last step in place on the second buffer), and odd and even larger counts (the final
    state ends in either buffer), eagerly and as a graph."""
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(B, 64, S, 64, generator=torch.Generator().manual_seed(5)).cuda()
    _say(f"chained ddim steps B={B} H={S} steps={steps}")
    old = os.environ.get("LDM_DDIMCHAIN")
    try:
        os.environ["LDM_DDIMCHAIN"] = "0"
        m0 = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
        plain = m0.ddim_sample(B, steps, spatial=(S, 64), z0=z0, use_graph=False)
        torch.cuda.synchronize()
    finally:
        if old is None:
            os.environ.pop("LDM_DDIMCHAIN", None)
        else:
            os.environ["LDM_DDIMCHAIN"] = old
    m1 = A.DiffusionModel(64, 3, 1000, 256, dtype="bf16", weights=w)
    eager = m1.ddim_sample(B, steps, spatial=(S, 64), z0=z0, use_graph=False)
    torch.cuda.synchronize()
    assert torch.isfinite(plain).all()
    d = (eager - plain).abs().max().item()
    _say(f"  max |chained - single| = {d:.3e}")
    assert torch.equal(eager, plain)
    for _ in range(2):
        graph = m1.ddim_sample(B, steps, spatial=(S, 64), z0=z0, use_graph=True)
        torch.cuda.synchronize()
        assert torch.equal(graph, plain)


@pytest.mark.parametrize("dtype,B,scaled", [("fp32", 4, False), ("fp32", 20, False), ("bf16", 20, False),
                                            ("bf16", 64, False), ("fp32", 3, True), ("bf16", 16, True)])
def test_whole_image_matrices_at_4x4_match_the_tap_kernels(A, O, dtype, B, scaled):
    """At the reference's 4x4 latents the sampler runs every spatial layer as one matrix over whole images.  Its
    per-step epsilon must agree with the tap-convolution path (noise_predict with LDM_DENSE4=0) on the sampler's
    own z_t, and with the oracle; batches that are not a multiple of the
    16 images the matrix path pads to, and the scaled UNet with its second attention (4 tokens), included."""
    cfg = A.LDMConfig.scaled() if scaled else A.LDMConfig()
    T, steps = 1000, 4
    w = A.make_synthetic_weights(cfg, 11)
    z0 = torch.randn(B, 64, 4, 4, generator=torch.Generator().manual_seed(B))
    m = _model(A, dtype, T, w, cfg)
    _say(f"dense4 {dtype} B={B} scaled={scaled}")
    got_z, zt, et = m.ddim_sample(B, steps, spatial=4, z0=z0.cuda(), use_graph=False, return_trace=True)
    torch.cuda.synchronize()
    ts = m.ddim_timesteps(steps)
    worst_tap, worst_ref = 0.0, 0.0
    for k in range(steps):
        temb = O.sin_time_addition(torch.full((B,), int(ts[k])), 256).cuda()
        with _tap_kernels_at_4x4():
            tap = m.noise_predict(zt[k], temb)
            torch.cuda.synchronize()
        worst_tap = max(worst_tap, rel_err(et[k], tap))
        worst_ref = max(worst_ref, rel_err(et[k], O.unet_forward(w, zt[k].cpu(), temb.cpu(), scaled_attention=scaled)))
    _say(f"dense4 {dtype} B={B}: eps vs tap kernels rel={worst_tap:.3e}, vs oracle rel={worst_ref:.3e}")
    assert worst_tap < (2e-5 if dtype == "fp32" else 2e-2)      # same arithmetic, another summation order
    assert worst_ref < TOL[dtype]
    zg = m.ddim_sample(B, steps, spatial=4, z0=z0.cuda(), use_graph=True)
    zg2 = m.ddim_sample(B, steps, spatial=4, z0=z0.cuda(), use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(zg, got_z) and torch.equal(zg2, got_z)


@pytest.mark.parametrize("dtype,B", [("fp32", 5), ("bf16", 37), ("bf16", 300)])
def test_noise_predict_at_4x4_with_a_time_per_image(A, O, dtype, B):
    """noise_predict on 4x4 latents (the training forward of the reference, src/main.rs:653-655, draws a timestep
    per image): the whole-image matrices with one time-vector row per image, against the tap kernels and the
    oracle.  B is not a multiple of 16; 300 images take the tiled 1x1 kernel, the others the skinny one."""
    w = A.make_synthetic_weights(A.LDMConfig(), 3)
    g = torch.Generator().manual_seed(B)
    z = torch.randn(B, 64, 4, 4, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    temb = O.sin_time_addition(t, 256)
    m = _model(A, dtype, 1000, w)
    _say(f"noise_predict 4x4 {dtype} B={B}")
    got = m.noise_predict(z.cuda(), temb.cuda())
    torch.cuda.synchronize()
    with _tap_kernels_at_4x4():
        tap = m.noise_predict(z.cuda(), temb.cuda())
        torch.cuda.synchronize()
    ref = O.unet_forward(w, z, temb)
    e_tap, e_ref = rel_err(got, tap), rel_err(got, ref)
    _say(f"noise_predict 4x4 {dtype} B={B}: vs tap kernels rel={e_tap:.3e}, vs oracle rel={e_ref:.3e}")
    assert e_tap < (2e-5 if dtype == "fp32" else 2e-2) and e_ref < TOL[dtype]
    assert e_tap > 0 or dtype == "fp32"       # the two paths really are different code


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_4x4_path_with_a_narrow_latent(A, O, dtype):
    """8 latent channels (padded to 64 inside): the whole-image matrices keep the padded columns zero, and the
    sampler's epsilon agrees with the oracle and with the tap kernels."""
    cfg = A.LDMConfig(8, 256, (64, 128, 256), False)
    w = A.make_synthetic_weights(cfg, 21)
    B, steps = 6, 3
    z0 = torch.randn(B, 8, 4, 4, generator=torch.Generator().manual_seed(5))
    m = _model(A, dtype, 1000, w, cfg)
    _say(f"narrow latent 4x4 {dtype}")
    got_z, zt, et = m.ddim_sample(B, steps, spatial=4, z0=z0.cuda(), use_graph=False, return_trace=True)
    torch.cuda.synchronize()
    assert tuple(got_z.shape) == (B, 8, 4, 4) and torch.isfinite(got_z).all()
    ts = m.ddim_timesteps(steps)
    for k in range(steps):
        temb = O.sin_time_addition(torch.full((B,), int(ts[k])), 256)
        ref = O.unet_forward(w, zt[k].cpu(), temb)
        with _tap_kernels_at_4x4():
            tap = m.noise_predict(zt[k], temb.cuda())
            torch.cuda.synchronize()
        assert rel_err(et[k], ref) < TOL[dtype]
        assert rel_err(et[k], tap) < (2e-5 if dtype == "fp32" else 2e-2)


def test_finiteness_probe_and_fallbacks(A, O):
    """The reference's probes outside its loop (src/main.rs:780-784, 788-792, 647-676), offered as an optional
    reduction after the loop: NaN / Inf injected into z0 propagate through the chain; check_finite=True replaces
    the result by zeros (latent, image) or by the sentinel loss 100, as the reference does; without it they
    propagate."""
    _say("finiteness probe")
    x = torch.randn(3, 64, 4, 4, generator=torch.Generator().manual_seed(1)).cuda()
    assert A.latent_is_finite(x)
    for bad, pos in ((float("nan"), 0), (float("inf"), 1500), (float("-inf"), x.numel() - 1)):
        y = x.clone()
        y.view(-1)[pos] = bad
        assert not A.latent_is_finite(y)
    odd = torch.randn(4099, generator=torch.Generator().manual_seed(2)).cuda()    # not a multiple of 4 elements
    assert A.latent_is_finite(odd)
    odd[4098] = float("nan")
    assert not A.latent_is_finite(odd)
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    wd = A.synthetic_from_shapes(A.decoder_param_shapes(3, 64), 21)
    m = A.DiffusionModel(64, 3, 10, 256, dtype="fp32", weights=w, decoder_weights=wd)
    z0 = torch.randn(2, 64, 4, 4, generator=torch.Generator().manual_seed(3)).cuda()
    good = m.denoising_process(2, spatial=4, z0=z0, seed=5, check_finite=True)
    assert torch.equal(good, m.denoising_process(2, spatial=4, z0=z0, seed=5)) and good.abs().max().item() > 0
    zbad = z0.clone()
    zbad[1, 7, 2, 2] = float("nan")
    assert not torch.isfinite(m.denoising_process(2, spatial=4, z0=zbad, seed=5)).all()
    fallback = m.denoising_process(2, spatial=4, z0=zbad, seed=5, check_finite=True)
    assert tuple(fallback.shape) == (2, 64, 4, 4) and (fallback == 0).all()
    assert (m.ddim_sample(2, 5, spatial=4, z0=zbad, check_finite=True) == 0).all()
    img = m.sample_images(2, z0=zbad, seed=5, check_finite=True)
    # the latent fallback is all zeros, so the image is decode(0): finite, and equal to the decoder's own output
    assert torch.isfinite(img).all() and torch.equal(img, m.decode(torch.zeros_like(z0)))
    # loss value: sentinel 100.0 when the latent is not finite, the ordinary value otherwise
    t = torch.tensor([1, 5]).cuda()
    n = torch.randn(2, 64, 4, 4, generator=torch.Generator().manual_seed(4)).cuda()
    assert m.forward(zbad, t, n, check_finite=True).item() == 100.0
    assert torch.equal(m.forward(z0, t, n, check_finite=True), m.forward(z0, t, n))


def test_alternating_samplers_on_one_model_keep_their_tables_and_graphs(A, O):
    """DDPM, DDIM (two step counts) and DDPM again on ONE sampler: every (mode, steps) pair keeps its own coefficient
    tables and captured graph, so each call must give what a fresh model gives for that call alone."""
    T, B, S = 40, 3, 8
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(21)).cuda()
    calls = [("ddpm", T), ("ddim", 10), ("ddim", 5), ("ddpm", T), ("ddim", 10), ("ddim", 5)]

    def run(model, mode, steps):
        if mode == "ddpm":
            return model.denoising_process(B, spatial=S, z0=z0, seed=9)
        return model.ddim_sample(B, steps, spatial=S, z0=z0)

    m = _model(A, "fp32", T, w)
    got = [run(m, *c) for c in calls]
    torch.cuda.synchronize()
    for c, g in zip(calls, got):
        fresh = run(_model(A, "fp32", T, w), *c)
        torch.cuda.synchronize()
        assert torch.equal(g, fresh), c
    assert torch.equal(got[0], got[3]) and torch.equal(got[1], got[4]) and not torch.equal(got[1], got[2])


def test_finiteness_probe_accepts_unaligned_views(A):
    """A view that starts inside a larger tensor (4-byte aligned only) is probed element by element at its ends."""
    base = torch.zeros(4099, device="cuda")
    for off in (0, 1, 2, 3, 5):
        for n in (1, 2, 3, 4, 7, 64, 4090):
            v = base[off:off + n]
            assert A.latent_is_finite(v), (off, n)
            for bad_at in {0, n - 1, n // 2}:
                base[off + bad_at] = float("inf")
                assert not A.latent_is_finite(v), (off, n, bad_at)
                base[off + bad_at] = 0.0
    base[0] = float("nan")
    assert A.latent_is_finite(base[1:]) and not A.latent_is_finite(base)


def test_bench_rccl_path_with_one_rank():
    """bench.py's multi-GPU path (nccl = RCCL process group, overlapped all_gather_into_tensor of the final latents,
    barrier, MAX all-reduce of the time) rehearsed on one GPU with a one-rank group; the line it prints must
    match the plain single-process run on the gathered batch size and be finite."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LDM_BENCH_FORCE_PG"] = "1"
    _say("bench.py with a one-rank RCCL group")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "8", "--spatial", "16", "--steps", "3",
                        "--warmup", "1", "--ddim-steps", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    r = json.loads(line[0])
    assert r["n_gpus"] == 1 and r["config"]["global_batch"] == 8 and r["value"] > 0
