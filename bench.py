#!/usr/bin/env python3
"""Headline benchmark: images/sec of a 50-step DDIM chain through the UNet on MI355X.

One "step" of this benchmark = one whole sampling call: 50 DDIM steps (strided over num_time = 1000)
of the default UNet (channels 64/128/256, latent 64) on a batch of 64 latents of 64x64 per GPU, bf16
activations/weights with f32 accumulation and f32 latent state, the whole loop replayed from one
hipGraph.  Inputs are synthetic and already resident in HBM; the VAE decode is outside the path.

Ranks.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N child processes itself (one
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before this process has touched the GPU, relays
rank 0's JSON line and exits with the children's status; under torch.distributed.run the ranks already
exist and are used as they are.  Every rank denoises its own chains and one all-gather collects the final
latents inside the timed region.  `--batch B` gives every GPU B chains (weak scaling, the default);
`--global-batch G` cuts G chains into contiguous slices over the ranks (strong scaling, BASELINE config 4).
Chain b of the global batch always starts from the same latent (counter-based generator keyed by the
global chain index), so the gathered result does not depend on the number of ranks.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline     -- the dominant kernel timed alone with HIP events on the launch stream.  bf16 / DDIM: the 3x3
                  tap convolution of the innermost level (six launches per UNet step), algorithmic FLOPs / average
                  duration against the dense bf16 MFMA peak; `traffic` = HBM bytes per launch from committed
                  rocprofv3 --pmc passes, looked up by the name of the kernel that the library really launched.
                  f32 / DDPM (config 5): the scheduler update, algorithmic bytes / average duration against HBM;
  cpu_baseline -- the CPU restatement of the reference's Rust path (oracle/, torch CPU f32) on the host cores:
                  config 1 (50-step DDPM, B = 4) at 64x64 and at the reference's 4x4 latents, one warm-up chain
                  and the median of three timed chains each (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak of MI355X
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def unet_flops(cfg_channels, cl, S, scaled, te=256):
    """2 * MACs of the UNet per sample (biases/activations excluded)."""
    c1, c2, c3 = cfg_channels
    P1, P2, P3 = S * S, S * S // 4, S * S // 16
    res = lambda C, P: 2 * 9 * C * C * P + te * C            # noqa: E731
    att = lambda C, P: 4 * C * C * P + 2 * P * P * C          # noqa: E731
    mac = 2 * 8 * te * te
    mac += 9 * cl * c1 * P1 + res(c1, P1)
    mac += 9 * c1 * c2 * P2 + res(c2, P2)
    mac += 9 * c2 * c3 * P3 + res(c3, P3)
    mac += res(c3, P3) * 2 + att(c3, P3)
    mac += 4 * c3 * c2 * P2 + 2 * c2 * c2 * P2 + res(c2, P2)
    if scaled:
        mac += att(c2, P2)
    mac += 4 * c2 * c1 * P1 + 2 * c1 * c1 * P1 + res(c1, P1)
    mac += 9 * c1 * cl * P1
    return 2.0 * mac


def _event_time(fn, iters):
    """Average duration (ms) of fn() over `iters` launches: HIP events on the stream the kernels are launched on."""
    import torch
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_dominant_kernel(dtype, B, S, C, iters=30):
    """3x3 tap convolution C->C on [B,S,S,C]: (average ms, FLOPs per launch, name of the kernel launched)."""
    import torch
    from diffusionmodel_amd import _lib
    from diffusionmodel_amd.ops import ConvOp
    g = torch.Generator().manual_seed(0)
    w = (torch.rand(C, C, 3, 3, generator=g) * 2 - 1) * (3.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.05
    op = ConvOp("conv3x3", w, b, dtype=dtype)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    x = torch.randn(B, S, S, C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(td)
    out = torch.empty_like(x)
    ms = _event_time(lambda: op(x, out=out), iters)
    name = _lib.load().ldm_last_kernel().decode()
    return ms, 2.0 * B * S * S * C * C * 9, name


def time_sched_kernel(B, S, C, iters=30):
    """The DDPM update on [B,C,S,S] f32 with the noise read from memory (src/main.rs:712-756): z read + write,
    eps read, noise read = 16 bytes per element.  (average ms, algorithmic bytes per launch)."""
    import torch
    import diffusionmodel_amd as A
    from diffusionmodel_amd import schedule
    gen = torch.Generator(device="cuda").manual_seed(2)
    z = torch.randn(B, C, S, S, device="cuda", generator=gen)
    eps = torch.randn(B, C, S, S, device="cuda", generator=gen)
    noise = torch.randn(B, C, S, S, device="cuda", generator=gen)
    row = schedule.ddpm_coefficients(1000)[500]
    ms = _event_time(lambda: A.sched_step(z, eps, row, "ddpm", last=False, noise=noise), iters)
    return ms, 16.0 * z.numel()


def pmc_traffic(kernel_name, shape):
    """HBM bytes per launch of `kernel_name` on `shape` from committed rocprofv3 --pmc passes (FETCH_SIZE x 2 on
    gfx950 + WRITE_SIZE, separate passes), or None when no record was taken on exactly this kernel and shape."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    for ent in rec.get("records", []):
        if ent.get("kernel") == kernel_name and ent.get("shape") == shape:
            return float(ent["hbm_bytes_per_launch"])
    return None


PMC_SQ_CSV = "r03_pmc_sq_per_kernel.csv"


def pmc_mfma_busy(kernel_name):
    """Share of the matrix pipes' cycles with an MFMA in flight for `kernel_name`, from the committed SQ counter passes
    (tools/pmc_sq.sh on the default workload's shapes: SQ_VALU_MFMA_BUSY_CYCLES over 4 SIMDs x 256 CUs x the dispatch's
    active cycles, and the same against 2.4 GHz x the traced duration), or None when that kernel is not in the file."""
    import csv
    short = kernel_name.replace("void ldm::", "").replace("ldm::", "").replace("(TapConvArgs)", "")
    try:
        with open(os.path.join(ROOT, "profiles", PMC_SQ_CSV)) as f:
            for r in csv.DictReader(f):
                if r["kernel"].replace("(TapConvArgs)", "") == short and r["mfma_busy"]:
                    return {"mfma_busy": float(r["mfma_busy"]), "mfma_busy_wall": float(r["mfma_busy_wall"] or 0.0),
                            "issue": float(r["issue"] or 0.0), "wait": float(r["wait"] or 0.0), "source": "profiles/" + PMC_SQ_CSV}
    except (OSError, ValueError, KeyError):
        return None
    return None


def cpu_baseline():
    """CPU restatement (oracle) on the host cores, BASELINE config 1: 50-step DDPM, B = 4, f32, at 64x64 latents and
    at the reference's own 4x4; one warm-up chain, then the median of three timed whole chains."""
    import torch
    from oracle import ldm_oracle as O
    import diffusionmodel_amd as A
    cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    w = A.make_synthetic_weights(A.LDMConfig(), 0)
    B, T = 4, 50
    out = {}
    with torch.no_grad():
        for S in (64, 4):
            z0 = torch.randn(B, 64, S, S, generator=torch.Generator().manual_seed(1234))
            noise = torch.randn(T, B, 64, S, S, generator=torch.Generator().manual_seed(4321))
            O.denoising_process(w, z0, T, noise=noise)                       # warm-up chain
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                O.denoising_process(w, z0, T, noise=noise)
                times.append(time.perf_counter() - t0)
            out[S] = statistics.median(times)
    return {"value": B / out[64], "unit": "images/s", "cores": cores, "kind": "port",
            "unet_steps_per_sec": T / out[64],
            "lat4_images_per_sec": B / out[4], "lat4_unet_steps_per_sec": T / out[4],
            "sample": f"whole 50-step DDPM chains, B=4, f32: 1 warm-up + median of 3 timed chains at [4,64,64,64] "
                      f"({out[64]:.2f} s per chain) and at the reference's [4,64,4,4] ({out[4]:.3f} s per chain); torch CPU "
                      f"restatement of src/main.rs (not the reference binary), {cores} threads"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, deadline_s=None):
    """Starts n fresh copies of this script, one per rank, and supervises them.  This process has not initialised the
    GPU (torch is not even imported yet), and it never replaces itself: children are ordinary subprocesses.
    The first rank that exits with an error ends the run: the others (which would wait for it in a collective for
    ever, holding their GPUs) are terminated, then killed, and its status is returned.  The same happens when the
    whole run exceeds `deadline_s` seconds (LDM_BENCH_DEADLINE_S, default 3000): status 124."""
    if deadline_s is None:
        deadline_s = float(os.environ.get("LDM_BENCH_DEADLINE_S", "3000"))
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    t_end = time.monotonic() + deadline_s
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1]
            print(f"bench: rank {bad[0][0]} exited with status {rc}; stopping the other ranks", file=sys.stderr, flush=True)
            break
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() > t_end:
            rc = 124
            print(f"bench: ranks still running after {deadline_s:.0f} s; stopping them", file=sys.stderr, flush=True)
            break
        time.sleep(0.05)
    live = [p for p in procs if p.poll() is None]
    for p in live:
        p.terminate()
    t_kill = time.monotonic() + 5.0
    for p in live:
        try:
            p.wait(timeout=max(0.0, t_kill - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc if rc > 0 else 1


def fake_sampler(local_batch, chain0, C=4, S=4):
    """CPU rehearsal of the rank logic (LDM_BENCH_FAKE=1): a deterministic function of the GLOBAL chain index."""
    import torch
    out = torch.empty(local_batch, C, S, S)
    for b in range(local_batch):
        out[b] = torch.randn(C, S, S, generator=torch.Generator().manual_seed(1000 + chain0 + b))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="chains per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="total chains, cut over the ranks (strong scaling)")
    ap.add_argument("--spatial", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ddpm"])
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--num-time", type=int, default=1000)
    ap.add_argument("--scaled", action="store_true", help="scaled UNet (128/256/512, attention at 2 resolutions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fake = os.environ.get("LDM_BENCH_FAKE") == "1"       # CPU rehearsal: no GPU, no library, gloo
    # LDM_BENCH_BACKEND=gloo: developer rehearsal of the multi-rank control flow with all ranks on one GPU
    # (host-staged gather); the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = "gloo" if fake else os.environ.get("LDM_BENCH_BACKEND", "nccl")
    # LDM_BENCH_FORCE_PG=1: build the process group (and take the gather / reduce path) even with one rank, so that a
    # one-GPU box can rehearse every RCCL call of the multi-GPU run
    use_pg = world > 1 or os.environ.get("LDM_BENCH_FORCE_PG") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            if not fake:
                torch.cuda.set_device(0)
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif not fake:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if fake and os.environ.get("LDM_BENCH_FAKE_FAIL_RANK") == str(rank):
        # rehearsal of a rank that dies after joining the group: the others then wait in their first collective
        sys.exit(3)

    import diffusionmodel_amd as A
    S = args.spatial
    if args.global_batch:
        global_batch = args.global_batch
        first, last = A.chain_range(global_batch, rank, world)
        B, chain0, scaling = last - first, first, "strong"
    else:
        B, chain0, scaling = args.batch, rank * args.batch, "weak"
        global_batch = B * world
    n_unet_steps = args.ddim_steps
    cfg = A.LDMConfig.scaled() if args.scaled else A.LDMConfig()
    num_time = args.num_time if args.sampler == "ddim" else args.ddim_steps
    sync = (lambda: None) if fake else torch.cuda.synchronize

    if fake:
        def sample():
            return fake_sampler(B, chain0)
    else:
        weights = A.make_synthetic_weights(cfg, 0)
        model = A.DiffusionModel(cfg.in_channels, 3, num_time, cfg.time_embed_dim, cfg.channels, cfg.scaled_attention,
                                 dtype=args.dtype, weights=weights)
        # initial latents of the global chains chain0 .. chain0 + B - 1: the counter-based generator, so chain b
        # starts from the same latent whatever the number of ranks
        z0 = A.philox_normal(B, cfg.in_channels, S, S, seed=1234, chain0=chain0)

        def sample():
            if args.sampler == "ddim":
                return model.ddim_sample(B, args.ddim_steps, spatial=S, z0=z0, chain0=chain0, use_graph=not args.no_graph)
            return model.denoising_process(B, spatial=S, z0=z0, seed=7, chain0=chain0, use_graph=not args.no_graph)

    pending = []      # all-gathers in flight (multi-GPU): call k's gather overlaps call k+1's chains

    def one_call():
        z = sample()
        if use_pg:
            pending.append(A.gather_latents_async(z, global_batch))
            while len(pending) > 2:           # bound the memory held by gathers in flight
                pending.pop(0).wait()
            return pending[-1]
        return z

    def fence():
        # every gather issued so far is complete when this returns: nothing leaves the timed region unfinished
        gathered = None
        while pending:
            gathered = pending.pop(0).wait()
        sync()
        if use_pg:
            dist.barrier()
            sync()
        return gathered

    for _ in range(args.warmup):
        out = one_call()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_call()
    gathered = fence()
    dt = time.perf_counter() - t0
    if use_pg:
        out = gathered
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all(), "non-finite latents"
    assert out.shape[0] == global_batch

    images_per_s = global_batch * args.steps / dt
    unet_steps_per_s = args.steps * n_unet_steps / dt           # per GPU (all ranks step in lock-step)
    flop_step = unet_flops(cfg.channels, cfg.in_channels, S, cfg.scaled_attention) * B
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS

    if rank == 0:
        res = {
            "metric": f"images/sec ({n_unet_steps}-step {args.sampler.upper()})",
            "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{n_unet_steps}-step {args.sampler.upper()} over num_time={num_time}, "
                                   f"{'scaled' if args.scaled else 'default'} UNet {list(cfg.channels)}, latent "
                                   f"[{B},{cfg.in_channels},{S},{S}] on rank 0, hipGraph={not args.no_graph}",
                       "global_batch": global_batch, "per_gpu_batch": B, "spatial": S,
                       "parallelism": f"dp{world} (batch-sharded chains, one all-gather of final latents)"},
            # what the process group itself reports (None without one): shows that RCCL saw every rank of an N > 1 run
            "collective_backend": dist.get_backend() if use_pg else None,
            "rccl_ranks": dist.get_world_size() if use_pg else None,
            "unet_steps_per_sec": unet_steps_per_s,
            "unet_tflops": flop_step * unet_steps_per_s / 1e12,
            "unet_frac_of_mfma_peak": flop_step * unet_steps_per_s / 1e12 / peak,
        }
        if fake:
            # rehearsal: a digest of the gathered tensor, to compare across numbers of ranks
            res["config"]["workload"] = "CPU rehearsal of the rank logic (LDM_BENCH_FAKE=1), nothing measured"
            res["gathered_digest"] = float(out.double().mul(torch.arange(1, out.numel() + 1).reshape(out.shape)).sum())
        else:
            # dominant kernel by share of the step (profiles/): the 3x3 tap convolution of the innermost
            # level (six launches per UNet step: down_res3, mid_res1, mid_res2 = 2 convs each)
            c_dom, s_dom = cfg.channels[2], S // 4
            kms, kflop, kname = time_dominant_kernel(args.dtype, B, s_dom, c_dom)
            achieved = kflop / (kms * 1e-3) / 1e12
            shape = f"3x3 {c_dom}->{c_dom} on [{B},{s_dom},{s_dom},{c_dom}] {args.dtype}"
            if S == 4:   # non-default run: at 4x4 latents the sampler runs whole-image matrices, not this kernel
                shape += " (tap kernel timed for reference; the 4x4 sampler uses the matrix path)"
            levels = {}
            for cc, ss in ((cfg.channels[0], S), (cfg.channels[1], S // 2)):
                ms_l, fl_l, name_l = time_dominant_kernel(args.dtype, B, ss, cc)
                levels[f"3x3 {cc}->{cc} on [{B},{ss},{ss},{cc}]"] = {"avg_ms": ms_l, "tflops": fl_l / (ms_l * 1e-3) / 1e12,
                                                                     "kernel": name_l}
            conv_roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": pmc_traffic(kname, shape),
                         "kernel": kname, "shape": shape, "avg_ms": kms, "flop_per_launch": kflop, "other_levels": levels}
            if args.dtype == "bf16" and B == 64 and S == 64 and not args.scaled:
                # counters are a separate, committed measurement (rocprofv3 --pmc cannot run inside this process):
                # taken on exactly this kernel and shape
                busy = pmc_mfma_busy(kname)
                if busy:
                    conv_roof.update(busy)
            if args.sampler == "ddpm" and args.dtype != "bf16":
                # config 5: the scheduler update is the HBM-bound kernel of the path
                sms, sbytes = time_sched_kernel(B, S, cfg.in_channels)
                gbs = sbytes / (sms * 1e-3) / 1e9
                res["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": gbs / PEAK_HBM_GBS, "traffic": pmc_traffic("ldm::sched_step_kernel", f"[{B},{cfg.in_channels},{S},{S}] f32 ddpm"),
                                   "kernel": "ldm::sched_step_kernel (DDPM update, noise read from memory)",
                                   "shape": f"[{B},{cfg.in_channels},{S},{S}] f32", "avg_ms": sms,
                                   "bytes_per_launch": sbytes, "convolution": conv_roof}
            else:
                res["roofline"] = conv_roof
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
