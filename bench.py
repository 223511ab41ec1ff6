#!/usr/bin/env python3
"""Headline benchmark: denoising steps/s of the DiffusionRenderer inverse pass on a 57 f x 576 x 1024 clip.

    python bench.py [--gpus N --steps K --warmup W]            (N=1 default)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one EDM Euler iteration of one G-buffer pass at guidance 0 (the node default, nodes.py:139): scale the
latent, ONE 7.2 B-parameter DiT forward over S = 18432 tokens, Euler update.  Inputs (noise latent, condition latent,
random-init weights of the real architecture) are synthetic and resident in HBM before the timed region.
With N > 1 the token rows of the ONE clip are sharded over the ranks (strong scaling): K/V all-gather over RCCL per
self-attention block, everything else token-local.

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline     - dominant kernel (by time) of the timed region, HIP-event timed on the launch stream
  cpu_baseline - the CPU oracle (oracle/dit_oracle.py, a bit-exact restatement of the reference's CPU path) timed on
                 this host's cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

TRAFFIC_PROFILE = "r03_pmc_traffic.json"
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


def dit_flops(S, D=4096, L=28, faithful=True):
    """Algorithmic FLOPs of one DiT forward: linear 28 D^2 L per token (+ attention 4 S D L).
    `faithful` counts the reference's dead cross-attention q/out projections; executed = without them (F8)."""
    lin = (28 if faithful else 24) * D * D * L * S
    att = 4.0 * S * S * D * L
    return lin + att


def host_cores() -> int:
    """CPU cores this process may actually use (affinity mask and cgroup quota, not the machine's core count)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:   # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    if "DRN_CPU_THREADS" in os.environ:
        return max(1, int(os.environ["DRN_CPU_THREADS"]))
    return min(n, 32)      # a 1-GPU share of the host; oversubscribing a shared box only slows the sample down


def cpu_baseline_full(pkg, net, sd_cpu, latent, steps):
    """BASELINE config 1 in full on the host cores: the whole 28-block model, `steps` Euler steps of
    the sampler loop (oracle/dit_oracle.py, the bit-exact CPU restatement of the reference), guidance 0."""
    from oracle import dit_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sw = pkg.synthetic_weights
    orc = O.DitOracle(sd_cpu, net, dtype=torch.bfloat16)
    F_, h, w = latent
    xT = (sw.synth_tensor("cpu.noise", (1, 16, F_, h, w), torch.float32, scale=1.7) * 80.0).to(torch.bfloat16)
    cond = sw.synth_tensor("cpu.cond", (1, 16, F_, h, w), torch.float32, scale=1.0).to(torch.bfloat16)
    ci = torch.full((1, 1), 3, dtype=torch.long)
    with torch.no_grad():
        orc.forward(xT, O.edm_sigmas(steps)[0], cond, ci)             # warm the thread pool / oneDNN primitives
        t0 = time.perf_counter()
        O.sample_loop(orc.forward, xT, cond, ci, steps, 0.0)
        dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"the whole workload: {steps} Euler steps of the {net['num_blocks']}-block model at S={F_ * (h // 2) * (w // 2)}, measured {dt:.1f}s"}


def cpu_baseline(pkg, S_full, budget_s=40.0):
    """Time the CPU oracle on one full-width block (FA+CA+MLP) and extrapolate x28 (a whole cfg-3 step is minutes)."""
    from oracle import dit_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    net = dict(pkg.diffusion_renderer_config.get_inverse_renderer_config()["net"], num_blocks=1)
    sd = pkg.synthetic_weights.synth_state_dict(net, torch.bfloat16)
    orc = O.DitOracle(sd, net, dtype=torch.bfloat16)

    def run(F_, h, w):
        x = pkg.synthetic_weights.synth_tensor("cpu.x", (1, 16, F_, h, w), torch.float32, scale=2.0).to(torch.bfloat16)
        c = pkg.synthetic_weights.synth_tensor("cpu.c", (1, 16, F_, h, w), torch.float32, scale=1.0).to(torch.bfloat16)
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.forward(x, torch.tensor(2.0), c, torch.full((1, 1), 3, dtype=torch.long))
        return time.perf_counter() - t0

    run(1, 16, 16)                                   # warm the thread pool / oneDNN primitives
    t_small = run(1, 128, 64)                        # S = 2048
    f_small, f_full = dit_flops(2048, L=1), dit_flops(S_full, L=1)
    est = t_small * f_full / f_small
    if est <= budget_s:
        t_blk = run(8, 72, 128) if S_full == 18432 else run(1, 2 * int((S_full) ** 0.5), 2 * int((S_full) ** 0.5))
        sample = f"1 of 28 blocks (+embed/final) at S={S_full}, measured {t_blk:.1f}s, x28"
    else:
        t_blk = est
        sample = (f"1 of 28 blocks at S=2048 measured {t_small:.2f}s, scaled by the FLOP ratio to S={S_full} "
                  f"({est:.0f}s/block est.), x28")
    return {"value": 1.0 / (28.0 * t_blk), "unit": "steps/s", "cores": cores, "kind": "port", "sample": sample}


CONFIGS = {   # BASELINE.json configs that fit one node: (frames, height, width, default steps, default warm-up)
    "cfg1": (1, 256, 256, 4, 1),       # 1 f 256^2: S = 256 tokens, a forward is a 14.5 GB weight stream (HBM roofline)
    "cfg2": (9, 512, 512, 5, 1),       # nearest legal clip to "8 frames 512^2" (SURVEY F11): latent (2,64,64), S = 2048
    "cfg2a": (8, 512, 512, 5, 1),      # the reference's own arithmetic for 8 frames: latent (1,64,64), S = 1024
    "cfg3": (57, 576, 1024, 3, 1),     # the headline: full Cosmos clip, S = 18 432
}
PROBE_FAILED_RC = 75          # exit code of a rank whose RCCL all-to-all probe failed (torchrun relays it as 1: see PROBE_MARKER)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


PROBE_MARKER = "[bench] PROBE_FAILED all_to_all_single"      # printed on stdout by a rank whose all-to-all probe failed


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` (how the driver calls it): start N ranks as a CHILD torch.distributed.run, relay rank 0's JSON
    line and the exit code.  Runs before anything touches the GPU; never an exec (gpurun forbids exec after GPU init).

    ONE situation starts a second run: a rank printed PROBE_MARKER, i.e. its all-to-all probe (tried before anything is timed)
    raised - this RCCL build cannot run the head <-> token exchange.  torch.distributed.run folds a rank's exit code into 1, so
    the marker line, not the code, identifies it.  A communicator is not reused after a failed collective, hence a FRESH run with
    DRN_SP_EXCHANGE=gather, and the relayed JSON line says so (`exchange_fallback`, `first_attempt_rc`).  Every other failure
    (a GPU fault, an assert, a kill) is relayed as it is, once, with the child's exit code; a child that outlives
    DRN_BENCH_TIMEOUT_S (default 3000) is terminated (SIGTERM, then SIGKILL) and the launcher exits 124.
    (A rendezvous port that was taken before torch.distributed.run could bind it - EADDRINUSE with no rank started - is not a run:
    the command is issued again on a fresh port, at most twice, `rendezvous_port_retries` in the relayed line.)"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("DRN_BENCH_TIMEOUT_S", "3000"))
    first_rc = None
    attempt, port_retries = 0, 0
    while attempt < 2:
        # (test hook, tests/test_bench_launcher.py: DRN_BENCH_FIRST_PORT = a port the test keeps busy, used for the first try only)
        hook_port = os.environ.get("DRN_BENCH_FIRST_PORT") if (attempt == 0 and port_retries == 0) else None
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", hook_port or str(free_port()), os.path.abspath(__file__)] + list(argv)
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        err_lines, out_lines = [], []

        def relay_stderr(pipe=proc.stderr, sink=err_lines):        # the ranks' stderr streams through as it comes (progress stays
            for ln in pipe:                                        # visible during a long run); a copy is kept for the port check
                sink.append(ln)
                sys.stderr.write(ln)
                sys.stderr.flush()

        def collect_stdout(pipe=proc.stdout, sink=out_lines):
            for ln in pipe:
                sink.append(ln)
        readers = [threading.Thread(target=relay_stderr, daemon=True), threading.Thread(target=collect_stdout, daemon=True)]
        for th in readers:
            th.start()
        try:
            proc.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            # torch.distributed.run puts every rank in a session of its own, so a group kill of the launcher's child would
            # orphan them: SIGTERM first (its handler terminates the ranks it started), SIGKILL only if it does not return
            proc.terminate()
            try:
                proc.wait(timeout=30)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)        # exactly the group this launcher started
                except ProcessLookupError:
                    pass
                proc.wait()
            print(f"[bench] {n}-rank run exceeded {limit:.0f} s: killed", file=sys.stderr, flush=True)
            return 124
        for th in readers:
            th.join(timeout=10)
        stdout, stderr = "".join(out_lines), "".join(err_lines)
        # the rendezvous port was taken between free_port() and torch.distributed.run's bind (no rank was started, nothing ran):
        # the same command again on another port, at most twice, and the relayed line says so
        if proc.returncode != 0 and not stdout.strip() and port_retries < 2 and \
                ("EADDRINUSE" in stderr or "address already in use" in stderr.lower()):
            port_retries += 1
            print(f"[bench] rendezvous port in use before any rank started; retry {port_retries} on a fresh port", file=sys.stderr, flush=True)
            continue
        line, probe_failed = None, False
        for ln in stdout.splitlines():
            t = ln.strip()
            if t.startswith("{") and '"metric"' in t:
                line = t
            elif t:
                probe_failed = probe_failed or PROBE_MARKER in t
                print(t, file=sys.stderr, flush=True)
        if line is not None and proc.returncode == 0:
            if attempt == 1 or port_retries:
                rec = json.loads(line)
                if attempt == 1:
                    rec["exchange_fallback"] = True
                    rec["first_attempt_rc"] = first_rc
                    rec["retried_exchange"] = "gather"
                if port_retries:
                    rec["rendezvous_port_retries"] = port_retries
                line = json.dumps(rec)
            print(line, flush=True)
            return 0
        if attempt == 0 and probe_failed and env.get("DRN_SP_EXCHANGE", "auto") != "gather":
            first_rc = proc.returncode
            print(f"[bench] all-to-all probe failed (rc {proc.returncode}); one fresh run with DRN_SP_EXCHANGE=gather", file=sys.stderr, flush=True)
            env["DRN_SP_EXCHANGE"] = "gather"
            attempt = 1
            continue
        return proc.returncode or 1
    return 1


def dry_run(args, world, rank):
    """Launcher rehearsal on CPU (tests/test_bench_launcher.py): gloo rendezvous at 127.0.0.1, a barrier, a MAX all-reduce and
    rank 0's JSON line - everything bench.py does around the timed region except the GPU work itself."""
    import torch.distributed as dist
    # test hook (tests/test_bench_launcher.py): DRN_DRYRUN_FAIL="<rank>:<rc>" makes that rank fail like a crashed child,
    # "<rank>:probe" like a rank whose all-to-all probe raised (only while the all-to-all exchange is selected, as the real probe)
    hook = os.environ.get("DRN_DRYRUN_FAIL", "")
    if hook:
        hr, what = hook.split(":")
        if int(hr) == rank:
            if what == "probe":
                if os.environ.get("DRN_SP_EXCHANGE", "auto") != "gather":
                    print(PROBE_MARKER + " (dry-run hook)", flush=True)
                    os._exit(PROBE_FAILED_RC)
            elif what == "hang":
                time.sleep(600)
            else:
                os._exit(int(what))
    if world > 1 or "WORLD_SIZE" in os.environ:
        dist.init_process_group("gloo")
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "denoising_steps_per_sec", "value": 0.0, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "rccl_ranks": world}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3", help="BASELINE.json config (default: the headline cfg3)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=28, help="debug only; the headline number needs 28")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tokenizer", action="store_true", help="skip the (untimed) tokenizer encode/decode leg")
    ap.add_argument("--no-cfg", action="store_true", help="skip the secondary guidance-2.0 (cond + uncond) figure")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal on CPU (gloo): no GPU work")
    ap.add_argument("--no-pass", action="store_true", help="skip the wall-clocked generate_video pass (pass_measured)")
    ap.add_argument("--pass-steps", type=int, default=35, help="denoising steps of the measured pass (BASELINE: 35)")
    ap.add_argument("--node-pass", action="store_true",
                    help="also wall-clock the inverse node's call: 5 G-buffer passes stepped as one batch (nodes.py:187-213)")
    args = ap.parse_args()
    cf, ch, cw, cs, cwu = CONFIGS[args.config]
    args.frames = cf if args.frames is None else args.frames
    args.height = ch if args.height is None else args.height
    args.width = cw if args.width is None else args.width
    args.steps = cs if args.steps is None else args.steps
    args.warmup = cwu if args.warmup is None else args.warmup

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # before any GPU call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, world, rank)
    # DRN_BENCH_BACKEND=gloo (rehearsal only, tests/test_parallel_gpu.py): the N ranks share THIS box's one GPU and exchange through
    # gloo's host staging - every line of the N > 1 path below runs (sharded engine, exchange timers, max-over-ranks timing, JSON
    # fields) except the RCCL transport itself.  The line is marked "transport": "gloo (rehearsal)" and is not a measurement.
    backend = os.environ.get("DRN_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("DRN_BENCH_BACKEND must be nccl or gloo")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or "WORLD_SIZE" in os.environ:          # under torch.distributed.run (also its 1-rank rehearsal)
        import torch.distributed as dist
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD
        probe = torch.zeros(world * 1024, dtype=torch.bfloat16, device=dev)
        if backend == "nccl":
            # bring the RCCL communicator (rings / xGMI peer mappings) up before any timed or warm-up step
            dist.all_gather_into_tensor(torch.empty(world * 1024, dtype=torch.bfloat16, device=dev), probe[:1024].contiguous())
            torch.cuda.synchronize()
        if backend == "nccl" and os.environ.get("DRN_SP_EXCHANGE", "auto") != "gather":
            # the exchange this run will use, tried once before anything is timed.  If this RCCL build cannot run it the
            # process group is NOT reused (a communicator is unreliable after a failed collective): every rank that sees
            # the failure exits with PROBE_FAILED_RC and the launcher starts a fresh run with the all-gather exchange
            try:
                dist.all_to_all_single(torch.empty_like(probe), probe)
                # ... and the uneven form the return exchange uses (parallel.alltoall_bands_: slabs for ranks < world - 1 first)
                if world > 1:
                    pr2 = probe.view(world, 1024)
                    pkg_par = load_package().parallel
                    back = torch.empty_like(pr2)
                    for lo, hi in ((0, world - 1), (world - 1, world)):
                        w_ = pkg_par.alltoall_bands_(pr2, back, lo, hi, pg, async_op=True)
                        if w_ is not None:
                            w_.wait()
                torch.cuda.synchronize()
            except Exception as e:                                       # noqa: BLE001 - any transport error
                print(f"[bench] rank {rank}: all_to_all_single failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
                print(PROBE_MARKER, flush=True)                # stdout: the launcher retries on this line only
                os._exit(PROBE_FAILED_RC)

    pkg = load_package()
    N = pkg.native
    N.load_library()
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config(args.height, args.width, args.frames)
    net = dict(cfg["net"], num_blocks=args.blocks)
    F_, h, w = (args.frames - 1) // 8 + 1, args.height // 8, args.width // 8
    S = F_ * (h // 2) * (w // 2)

    sw = pkg.synthetic_weights
    sd = sw.synth_state_dict(net, torch.bfloat16, device=dev)
    model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(dict(cfg, net=net), device=dev, process_group=pg)
    model.load_state_dict(sd, strict=True)
    # small clips: the CPU baseline runs the WHOLE workload (4 steps of the full model), on a host copy of the same weights
    full_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline and S <= 256 and args.blocks == 28)
    sd_cpu = {k: v.cpu() for k, v in sd.items()} if full_cpu else None
    del sd
    torch.cuda.empty_cache()

    total = args.steps + args.warmup
    model.scheduler.set_timesteps(max(total, 2))
    sig = model.scheduler.sigmas
    xt = (sw.synth_tensor("bench.noise", (1, 16, F_, h, w), torch.float32, device=dev, scale=1.7) * sig[0].item()).to(torch.bfloat16)
    cond = sw.synth_tensor("bench.cond", (1, 16, F_, h, w), torch.float32, device=dev, scale=1.0).to(torch.bfloat16)

    def one_step(i, x):
        t = model.scheduler.timesteps[i]
        xs = model.scheduler.scale_model_input(x, timestep=t)
        out = model.net(x=xs, timesteps=t, latent_condition=cond, context_index=3)
        model.scheduler.current_step = i
        return model.scheduler.step(out, t, x).prev_sample

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        xt = one_step(i, xt)
    barrier()
    # HIP-event pairs around every 7th GEMM / attention launch (7 is coprime with the 4-GEMM-per-block pattern, so all four
    # shapes are sampled evenly); bracketing every launch would add ~10 ms of queue time per step
    # (small shapes replay the forward as one hipGraph, where per-launch events do not apply: no roofline there)
    # (sharded runs: every 29th launch, the event pairs sit on rank 0 only and would skew its step against the other ranks)
    timer = N.KernelTimer(sample_every=7 if world == 1 else 29) if (rank == 0 and S > 4096 and not os.environ.get("DRN_NO_TIMER")) else None
    N.set_timer(timer)
    xtimer = pkg.parallel.ExchangeTimer() if (pg is not None and rank == 0) else None      # (also the 1-rank RCCL rehearsal)
    pkg.parallel.set_exchange_timer(xtimer)
    if pg is not None and world == 1:
        pkg.parallel.SINGLE_RANK_COLLECTIVES = True       # 1-rank rehearsal under torchrun: issue the RCCL calls anyway
    host_loop = 0.0
    t0 = time.perf_counter()
    # inside the timed region: the timed steps' AdaLN vectors, batched as generate_samples_from_batch does before its loop
    model.net.prepare_timesteps([float(model.scheduler.timesteps[i]) for i in range(args.warmup, total)])
    for i in range(args.warmup, total):
        h0 = time.perf_counter()
        xt = one_step(i, xt)
        host_loop += time.perf_counter() - h0         # wall time of the host loop: enqueue work + queue back-pressure
    barrier()
    elapsed = time.perf_counter() - t0
    N.set_timer(None)
    pkg.parallel.set_exchange_timer(None)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = tmax.item()
    assert torch.isfinite(xt.float()).all(), "non-finite latent"
    # the host's own cost of a step: ONE step enqueued on an EMPTY queue (nothing to wait for), outside the timed region.  The
    # loop figure above also contains the time the host spent blocked on the full HIP queue, i.e. mostly GPU time.
    model.scheduler.set_timesteps(max(total, 2))
    barrier()
    h0 = time.perf_counter()
    _x = one_step(0, xt)
    host_empty_ms = 1e3 * (time.perf_counter() - h0)
    barrier()
    del _x

    # ---- secondary figure: the same step at the pipeline's default guidance 2.0 = cond + uncond forwards, which the
    #      sampler runs as ONE batch of two clips; one warm-up + two timed steps, outside the headline's timed region
    cfg_ms = None
    if not args.no_cfg and world == 1:          # (sharded runs: headline only, nothing extra that could cost the line)
        def cfg_step(i, x):
            t = model.scheduler.timesteps[i]
            xs = model.scheduler.scale_model_input(x, timestep=t)
            both = model.net(x=torch.cat([xs, xs], 0), timesteps=t, latent_condition=torch.cat([cond, torch.zeros_like(cond)], 0),
                             context_index=[3, 0])
            out = N.cfg_combine(both[:1].contiguous(), both[1:].contiguous(), 2.0)
            model.scheduler.current_step = i
            return model.scheduler.step(out, t, x).prev_sample
        xg = cfg_step(0, xt)
        barrier()
        t0 = time.perf_counter()
        for i in (0, 1):
            xg = cfg_step(i, xg)
        barrier()
        cfg_ms = (time.perf_counter() - t0) / 2 * 1e3
        del xg

    # ---- tokenizer leg (rank 0, outside the timed region): one encode + one decode of the full clip, for frames/s and the
    #      conv kernel's achieved HBM rate.  Random-init CV8x8x8 weights; synthetic RGB clip resident in HBM.
    tok, vae = None, None
    if rank == 0 and not args.no_tokenizer:
        try:
            vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=dev), device=dev)
            clip = sw.synth_tensor("bench.rgb", (1, 3, args.frames, args.height, args.width), torch.float32, device=dev).to(torch.bfloat16)
            vae.decode(vae.encode(clip))                   # warm-up (allocator, code objects)
            torch.cuda.synchronize()
            vt = N.KernelTimer(names=("conv",))
            N.set_timer(vt)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            z = vae.encode(clip)
            e[1].record()
            rec = vae.decode(z)
            e[2].record()
            torch.cuda.synchronize()
            N.set_timer(None)
            u8 = N.postprocess_u8(rec.contiguous(), False)
            assert u8.shape == (1, args.frames, args.height, args.width, 3)
            cs = vt.summary()["conv"]
            tok = {"encode_ms": round(e[0].elapsed_time(e[1]), 2), "decode_ms": round(e[1].elapsed_time(e[2]), 2),
                   "conv_launches": cs["launches"], "conv_ms": round(cs["ms_total"], 2),
                   "conv_tflops": round(cs["flops"] / (cs["ms_total"] * 1e-3) / 1e12, 1),
                   "conv_hbm_gbs_algorithmic": round(cs["bytes"] / (cs["ms_total"] * 1e-3) / 1e9, 1),
                   "hbm_frac": round(cs["bytes"] / (cs["ms_total"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            del clip, z, rec, u8
            torch.cuda.empty_cache()
        except Exception as e:                      # noqa: BLE001 - the headline line must survive a failure of this optional leg
            print(f"[bench] tokenizer leg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            tok = None
            vae = None

    # ---- a REAL pass through the drop-in boundary (rank 0, one GPU, outside the headline's timed region): the reference's
    #      CleanDiffusionRendererPipeline.generate_video (diffusion_renderer_pipeline.py:242-321) on a HOST clip - H2D, tokenizer
    #      encode, condition assembly, prepare_timesteps, N Euler steps at guidance 0, decode, post-process, D2H - wall-clocked
    #      from the call to the uint8 array.  frames/s of BASELINE.json's metric = frames / that time.
    pass_measured, node_measured = None, None
    if rank == 0 and world == 1 and tok is not None and not args.no_pass and args.blocks == 28:
        try:
            pipe = pkg.diffusion_renderer_pipeline.CleanDiffusionRendererPipeline(
                "/nonexistent", "synthetic.pt", model_type=None, vae_instance=vae, model_instance=model, guidance=0.0,
                num_steps=args.pass_steps, seed=42)
            pipe.set_model_type("inverse")
            rgb = sw.synth_tensor("bench.rgb", (1, 3, args.frames, args.height, args.width), torch.float32)     # host, fp32 (IMAGE)
            batch = {"rgb": rgb, "video": rgb, "context_index": torch.full((1, 1), 3, dtype=torch.long)}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            u8 = pipe.generate_video(batch, normalize_normal=True, seed=42)
            dt = time.perf_counter() - t0
            assert u8.shape == (1, args.frames, args.height, args.width, 3) and u8.dtype.name == "uint8"
            pass_measured = {"seconds": round(dt, 3), "frames_per_sec": round(args.frames / dt, 3), "steps": args.pass_steps,
                             "what": "generate_video(host fp32 clip) -> uint8 ndarray: H2D + encode + conditions + "
                                     f"{args.pass_steps} steps (guidance 0) + decode + post-process + D2H, wall clock"}
            del u8
            if args.node_pass:
                node = pkg.nodes.Cosmos1InverseRenderer()
                image = ((rgb + 1.0) * 0.5).permute(0, 2, 3, 4, 1).contiguous()        # (B,T,H,W,C) in [0,1], as ComfyUI hands it over
                pipe._h2d_cache = {}
                model._enc_cache.clear()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                outs = node.run_inverse_pass(pipe, image, guidance=0.0, seed=42)
                dt5 = time.perf_counter() - t0
                assert len(outs) == 5 and tuple(outs[0].shape) == (args.frames, args.height, args.width, 3)
                node_measured = {"seconds": round(dt5, 3), "passes": 5, "frames_per_sec_per_pass": round(5 * args.frames / dt5, 3),
                                 "steps": args.pass_steps,
                                 "what": "Cosmos1InverseRenderer.run_inverse_pass: 5 G-buffer passes stepped as one batch, one "
                                         "encode, 5 decodes, uint8 -> float IMAGE tensors on the host"}
                del outs
            del pipe, rgb, batch
        except Exception as e:                      # noqa: BLE001 - optional leg
            print(f"[bench] measured pass failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    vae = None
    torch.cuda.empty_cache()

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        steps_s = args.steps / elapsed
        fl_faithful, fl_exec = dit_flops(S, L=args.blocks), dit_flops(S, L=args.blocks, faithful=False)
        summ = timer.summary() if timer is not None else {"gemm": {"launches": 1, "ms_total": 1.0, "ms_avg": 1.0, "flops": 0.0, "bytes": 0.0}}
        dom = max(summ, key=lambda k: summ[k]["ms_total"])
        d = summ[dom]
        achieved = d["flops"] / (d["ms_total"] * 1e-3) / 1e12        # TFLOP/s over that kernel's launches
        # L2-miss traffic per launch of that kernel family from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per MI355X_MICROARCH.md); null if the profile is absent
        traffic = None
        try:
            if S != 18432 or args.blocks != 28 or world != 1:
                raise KeyError("profile was taken on the headline configuration only")
            with open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)) as f:
                traffic = json.load(f)["families"][dom]["bytes_per_launch_corrected"]
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "mfma", "kernel": {"gemm": "gemm256s_kernel" if world == 1 else "gemm144_kernel", "attention": "attention16_fwd_kernel" if os.environ.get("DRN_ATT16", "1") != "0" else "attention_fwd_kernel"}[dom],
                    "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "bytes/launch (L2-miss traffic incl. Infinity-Cache hits)",
                    "traffic_source": f"profiles/{TRAFFIC_PROFILE}: committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command, NOT "
                                      "measured in this run (goes stale when a kernel changes)",
                    "launches_per_step": d.get("launches_seen", d["launches"]) // args.steps, "launches_timed": d["launches"],
                    "avg_launch_ms": round(d["ms_avg"], 4),
                    "per_kernel": {k: {"tflops": round(v["flops"] / (v["ms_total"] * 1e-3) / 1e12, 1),
                                       "ms_per_step": round(v["ms_avg"] * v.get("launches_seen", v["launches"]) / args.steps, 2)}
                                   for k, v in summ.items()}}
        if timer is None:
            # small clips (S <= 4096): a forward is a weight stream (HBM roofline); per-launch event pairs would cost
            # ~9 % of a 7 ms step, so the rate is the executed linears' weight bytes over the WHOLE step (a lower bound on the
            # GEMM kernels' own rate)
            wbytes = 12.0 * 4096 * 4096 * 2 * args.blocks
            gbs = wbytes / (ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "few-token linears (gemm_tall_kernel, gemm256w_kernel slices)", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                        "note": "weight bytes of the executed linears / whole step time (no per-launch events at this size)"}
            if S >= 1024:
                # config 2 (S = 1024 / 2048): 256+ FLOP per weight byte x 4-8 -> the MFMA roofline; still no
                # per-launch events (a pair costs ~35 us, 3 % of a 24 ms step): executed FLOPs over the whole step
                tf = fl_exec / (ms * 1e-3) / 1e12
                roofline = {"bound": "mfma", "kernel": "whole DiT step (GEMM tile kernels + attention_fwd_kernel)",
                            "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / PEAK_BF16_TFLOPS, 4), "traffic": None,
                            "note": "executed FLOPs / whole step time (no per-launch events at this size): a lower bound on "
                                    "the MFMA kernels' own rate"}
        out = {
            "metric": "denoising_steps_per_sec", "value": round(steps_s, 4), "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "rccl_ranks": world,
            **({"transport": "gloo (rehearsal: ranks share one GPU, host-staged exchanges - not a measurement)"} if backend == "gloo" else {}),
            "host_enqueue_ms_empty_queue": round(host_empty_ms, 2),
            "host_loop_wall_ms_per_step": round(1e3 * host_loop / args.steps, 2),
            "config": {"workload": f"inverse pass, {args.frames}f x {args.height} x {args.width} clip: EDM Euler step = 1 DiT "
                                   f"forward (D=4096, {args.blocks} blocks, 32 heads) over S={S} tokens, guidance 0",
                       "latent": [16, F_, h, w], "tokens": S, "parallelism": (f"sp{world} (token bands; self-attention exchange: "
                                       + {"none": "none", "a2a": "head<->token all-to-all", "gather": "K/V all-gather"}[model.net.exchange] + ")"),
                       "weights": "random-init (hash generator), 7.2e9 params bf16"},
            "dit_tflops_reference_equivalent": round(fl_faithful / (ms * 1e-3) / 1e12, 1),
            "dit_tflops_executed": round(fl_exec / (ms * 1e-3) / 1e12, 1),
            "mfma_frac_executed": round(fl_exec / (ms * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            "frames_per_sec_35step_pass_dit_only": round(args.frames / (35 * ms * 1e-3), 3),
            "roofline": roofline,
        }
        if xtimer is not None:
            # rank 0's view: time its compute stream waited in front of each exchange, per self-attention layer
            xs = xtimer.summary()
            out["exchange"] = {"kind": model.net.exchange, "exposed_ms_per_layer": {k: round(v["ms_avg"], 4) for k, v in xs.items()},
                               "exposed_ms_per_step": round(sum(v["ms_avg"] * v["waits"] for v in xs.values()) / args.steps, 3),
                               "waits_timed": {k: f'{v["timed"]} of {v["waits"]}' for k, v in xs.items()}}
        if cfg_ms is not None:
            out["guidance_2"] = {"steps_per_sec": round(1e3 / cfg_ms, 4), "ms_per_step": round(cfg_ms, 2),
                                 "note": "cond + uncond forwards of one Euler step as one batch of two clips (pipeline default guidance)"}
        if tok is not None:
            out["tokenizer"] = tok
            out["frames_per_sec_35step_pass"] = round(args.frames / ((tok["encode_ms"] + 35 * ms + tok["decode_ms"]) * 1e-3), 3)
            out["frames_per_sec_35step_pass_note"] = "composed: frames / (encode + 35 x ms_per_step + decode); pass_measured is the wall clock"
        if pass_measured is not None:
            out["pass_measured"] = pass_measured
            if tok is not None:
                comp = (tok["encode_ms"] + args.pass_steps * ms + tok["decode_ms"]) * 1e-3
                pass_measured["composed_seconds"] = round(comp, 3)
                pass_measured["measured_over_composed"] = round(pass_measured["seconds"] / comp, 4)
        if node_measured is not None:
            out["node_pass_measured"] = node_measured
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_full(pkg, net, sd_cpu, (F_, h, w), 4) if sd_cpu is not None else cpu_baseline(pkg, S)
        print(json.dumps(out), flush=True)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
