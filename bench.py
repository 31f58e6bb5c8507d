#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec (rollout + update), pH env, 16 384 parallel envs per MI355X.

One "step" = one pass of the hot path over one batch of synthetic input:
  rollout : 16 384 lanes x one 50-step episode of 'PH1DChangingParamUniformGoalIntegrator-SqaureDistance-v35'
            (ensemble params resampled every episode, in-kernel Philox draws) under the residual modular PPO
            policy (net_dim 128): fused f32-MFMA policy forward -> exploration noise -> fused residual env step
  update  : value pass (fused f32-MFMA critic forward over 819 200 rows) -> GAE scan -> PPO minibatch updates,
            batch 65 536, repeat_times 8 -> 100 optimizer steps (run_ph_changing.sh:5,9 scaled to N*T samples)
  => 819 200 env-steps per step and per GPU.  Weak scaling: every rank owns 16 384 lanes; one flat-gradient
  all-reduce per optimizer step.

Usage:  python bench.py [--gpus N --steps K --warmup W]
  N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE /
  MASTER_* from the env), and as plain `python bench.py --gpus N`: the parent then starts N rank processes itself BEFORE it
  touches the GPU (it never initialises HIP, never re-execs), relays rank 0's JSON line and exits non-zero if a rank fails.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

LANES = 16384
T_EP = 50
NET_DIM = 128
BATCH = 65536
REPEAT = 8
LAMBDA = 0.99     # run_ph_changing.sh:10
GAMMA = 0.99      # train.py never forwards --gamma

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD

# algorithmic cost models (DESIGN.md "Kernels")
PH_STEP_BYTES = 97            # mixed mode, fused-residual step: 64 B read + 33 B written per env-step
MLP_FLOPS_PER_ROW = {"critic": 2 * (3 * 128 + 128 * 128 * 2 + 128),
                     "modular_actor": 2 * (2 * 128 + 128 * 64 + 1 * 128 + 128 * 64 + 128 * 128 + 128)}
# one minibatch gradient = forward + backward-dX + dW of both nets: 3 x 2 flop per weight per sample
GRAD_FLOPS_PER_SAMPLE = 3 * (MLP_FLOPS_PER_ROW["critic"] + MLP_FLOPS_PER_ROW["modular_actor"])


class KernelTimer:
    """HIP-event timing of named launches on torch's current stream (the stream libpime_hip launches on)."""

    def __init__(self):
        self.pairs = {}
        self.enabled = False

    def wrap(self, name, fn):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.pairs.setdefault(name, []).append((s, e))
            return out
        return timed

    def bracket(self, name, thunk):
        if not self.enabled:
            return thunk()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = thunk()
        e.record()
        self.pairs.setdefault(name, []).append((s, e))
        return out

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(s.elapsed_time(e) for s, e in v) / len(v)) for k, v in self.pairs.items()}


def pmc_traffic_bytes(kernel_prefixes, pattern="r*_pmc_hbm_traffic.json"):
    """(HBM bytes per launch, source file) from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    bench (profiles/r<round>_<tag>_pmc_hbm_traffic.json, produced by tools/pmc_traffic.sh: separate passes, FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950; PIME_PMC_TRAFFIC=<file> names another one).  PMC counters cannot be read
    from inside the process, so this is the measured figure of the same command on the same kernels -- the file is named in
    the JSON line so a stale one shows -- or (None, None)."""
    import glob
    path = os.environ.get("PIME_PMC_TRAFFIC")
    if not path:
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        path = found[-1] if found else None
    if not path or not os.path.exists(path):
        return None, None
    data = json.load(open(path))
    total = 0.0
    for pref in kernel_prefixes:   # every kernel of the profile whose name contains the prefix (one launch of each per minibatch)
        hit = [v for k, v in data.items() if pref in k]
        if not hit:
            return None, os.path.relpath(path, ROOT)
        total += sum((h["fetch_mb_corrected"] + h["write_mb"]) * 1024 * 1024 for h in hit)
    return total, os.path.relpath(path, ROOT)


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def build_stack(device, rank, world, dp):
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.run import make_buffer
    env = gym_control.make_vec(gym_control.PH_V35, LANES, device=device, state_mode="mixed", seed=0,
                               env_offset=rank * LANES, draws="philox", resample_every=1)
    torch.manual_seed(0)  # identical initial replicas on every rank
    agent = AgentResidualIntegratorModularPPO(device=device)
    agent.lambda_gae_adv = LAMBDA
    agent.init(NET_DIM, env.state_dim, 1, env.n_integrator)
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    agent.init_actor_zero()
    agent.fix_K()
    agent.dp = dp
    if dp is not None:
        dp.broadcast_module(agent.act, agent.cri)
    torch.manual_seed(1000 + rank)  # exploration / minibatch streams differ per rank
    buf = make_buffer(agent, env, LANES * T_EP)
    return env, agent, buf


def one_step(env, agent, buf):
    steps = agent.explore_env(env, buf, LANES * T_EP, 1.0, GAMMA)
    agent.update_net(buf, LANES * T_EP, BATCH, REPEAT)
    return steps


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The same hot path on the host cores of this box: the env on libpime_cpu.so -- the CPU twin of the env entry
    points, i.e. the product's own lane functions compiled for the host (include/pime_cpu.h), threads over the
    lanes -- + torch-CPU nets driven by the product's own agent code, on a bounded sample: ONE step of the same workload
    (16 384 lanes x one 50-step episode, batch 65 536, repeat 8 -> 100 optimizer steps) on all cores -> `value`; plus
    env-only points (prior controller + tanh(N(0,1) e^-0.5) residual) at N = 1 / 4 096 / 16 384 with one
    thread and with all cores."""
    import oracle  # noqa: F401  (allowed here: bench.py's cpu_baseline leg)
    from oracle.cpu_stack import OracleBackend, TwinVecEnv
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    # the box's CPU share, not the host's core count (oversubscribing torch's intra-op pool stalls for minutes)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    oracle.set_threads(cores)
    n = LANES  # the full workload
    env = TwinVecEnv("ph", n, seed=0, threads=cores)
    torch.manual_seed(0)
    agent = AgentResidualIntegratorModularPPO(backend=OracleBackend(), device="cpu")
    agent.lambda_gae_adv = LAMBDA
    agent.init(NET_DIM, 3, 1, 1)
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    buf = TrajectoryBuffer(T_EP, n, 3, 1, "cpu")
    t0 = time.perf_counter()
    steps = agent.explore_env(env, buf, n * T_EP, 1.0, GAMMA)
    t1 = time.perf_counter()
    agent.update_net(buf, n * T_EP, n * T_EP * BATCH // (LANES * T_EP), REPEAT)
    t2 = time.perf_counter()
    # env-only points: reset + 50 steps (+ auto-reset / resample) per episode under prior + exploration-noise residual
    points = []
    rng = np.random.RandomState(0)
    K = -env.K
    for lanes in (1, 4096, 16384):
        for thr in (1, cores):
            oracle.set_threads(thr)
            e = oracle.OraclePH(lanes, env.table, seed=0)
            obs = e.reset()
            episodes = max(1, min(2000, int(4e6 // (lanes * T_EP))))   # ~4M env-steps at most per point
            a_pre = (rng.standard_normal((T_EP, lanes)) * np.exp(-0.5)).astype(np.float32)
            ta = time.perf_counter()
            for _ in range(episodes):
                for t in range(T_EP):
                    act = oracle.residual_action(a_pre[t], obs, K)
                    obs, _, _, _ = e.step(act, auto_reset=True)
            tb = time.perf_counter()
            points.append({"lanes": lanes, "threads": thr, "env_steps_per_s": episodes * T_EP * lanes / (tb - ta)})
            if lanes == 1:
                break   # one lane has nothing to spread over threads
    oracle.set_threads(cores)
    from oracle.twin import TwinEnv
    for lanes in (4096, 16384):   # the same sweep on the CPU twin (product arithmetic on the host)
        for thr in (1, cores):
            e = TwinEnv("ph", lanes, table=env.table, seed=0, threads=thr)
            obs = e.reset()
            episodes = max(1, min(2000, int(4e6 // (lanes * T_EP))))
            a_pre = (rng.standard_normal((T_EP, lanes)) * np.exp(-0.5)).astype(np.float32)
            ta = time.perf_counter()
            for _ in range(episodes):
                for t in range(T_EP):
                    obs, _, _ = e.step_residual(a_pre[t], obs, K)
            tb = time.perf_counter()
            points.append({"lanes": lanes, "threads": thr, "env": "twin", "env_steps_per_s": episodes * T_EP * lanes / (tb - ta)})
            e.close()
    return {"value": steps / (t2 - t0), "unit": "env-steps/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "nproc": os.cpu_count(),
            "env": "libpime_cpu.so: the product's lane functions (csrc/env_device.hpp) compiled for the host, float64 state",
            "sample": f"{n} lanes x {T_EP} steps (= {steps} env-steps), batch {n * T_EP * BATCH // (LANES * T_EP)}, "
                      f"repeat {REPEAT}: CPU-twin env + torch-CPU policy forward ({cores} threads) {t1 - t0:.2f}s "
                      f"+ torch-CPU PPO update ({cores} threads) {t2 - t1:.2f}s",
            "env_only_points": points,
            "reference_python_n1": {"value": 316, "unit": "env-steps/s", "where": "build container, 8 threads (BASELINE.md §2); "
                                    "the reference cannot travel to the GPU box"}}


def emit(json_fd, out):
    """The ONE JSON line.  Opt-in kernel variants (environment switches read by libpime_hip.so) are named in it."""
    variants = []
    if os.environ.get("PIME_MLP16"):
        variants.append("PIME_MLP16=1: widths 64 / 128 on the streamed 16-tile kernel family")
    if os.environ.get("PIME_GRAD_BF16X3", "0") not in ("", "0"):
        variants.append("PIME_GRAD_BF16X3=1: the streamed layers of the 16-tile gradient kernels (widths 128 / 256) as six "
                        "v_mfma_f32_16x16x32_bf16 per product block, every f32 operand split into three bf16 pieces (f32-level "
                        "error; roofline.achieved stays f32-equivalent flops against the f32 matrix peak)")
        out["dtype"] = "f32 (3xbf16 split operands, f32 accumulate)"
    if variants:
        out.setdefault("config", {})["variant"] = "; ".join(variants)
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def grad_roofline(agent, update, D, md, kernel):
    """`roofline` block of a PPO workload: HIP events around the gradient launches of every minibatch of ONE extra, untimed update
    (agent.launch_timer; the first such update captures the two-graph step sequence and is discarded), against the f32 matrix peak.
    Algorithmic flops per sample: forward + input-gradient chain + weight gradients (3x) of both nets, each 2 (D md + 2 md^2 + md)
    multiply-adds (the modular actor's towers add up to the same count)."""
    timer = KernelTimer()
    timer.enabled = True
    agent.launch_timer = timer.bracket
    update()
    timer.pairs.clear()
    update()
    torch.cuda.synchronize()
    agent.launch_timer = None
    n_g, ms_g = timer.summary()["ppo_minibatch_grad"]
    flops = 2 * 3 * 2 * (D * md + 2 * md * md + md) * BATCH
    tf = flops / (ms_g * 1e-3) / 1e12
    out = {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": tf / F32_MFMA_PEAK_TFLOPS, "traffic": None, "launches_per_step": n_g, "avg_launch_ms": ms_g,
           "algorithmic_flops_per_launch": flops}
    share = bf16x3_share(md)
    if share > 0:   # the opt-in variant: its own peak = what six bf16 MFMAs per f32 product allow on the share of the flops that take them
        peak = 1.0 / (share / (BF16_MFMA_PEAK_TFLOPS / 6.0) + (1.0 - share) / F32_MFMA_PEAK_TFLOPS)
        out.update({"peak": peak, "frac": tf / peak, "frac_of_f32_peak": tf / F32_MFMA_PEAK_TFLOPS,
                    "peak_note": f"bf16x3: {share:.2f} of the flops as six v_mfma_f32_16x16x32_bf16 per product block "
                                 f"({BF16_MFMA_PEAK_TFLOPS:.0f} / 6 = {BF16_MFMA_PEAK_TFLOPS / 6:.1f} TFLOP/s f32-equivalent), the rest f32 MFMA "
                                 f"({F32_MFMA_PEAK_TFLOPS}); achieved counts f32-equivalent flops"})
    return out


BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak (MI355X_MICROARCH.md)


def bf16x3_share(md):
    """Share of a minibatch gradient's flops that the opt-in PIME_GRAD_BF16X3=1 variant runs on bf16 matrix instructions: the streamed
    chain layers (2/3: forward and dX) at widths 128 / 256 of the 16-tile family, plus the weight gradients at width 256."""
    if os.environ.get("PIME_GRAD_BF16X3", "0") in ("", "0"):
        return 0.0
    if md == 256:
        return 1.0
    return 2.0 / 3.0 if md == 128 and os.environ.get("PIME_MLP16") else 0.0


def bench_water_tank(args, device, json_fd):
    """BASELINE config 2: water-tank Integrator env, 4096 lanes x 200-step episodes, ResidualIntegratorModularPPO
    net_dim 128, batch 65536, repeat 8 (run_watertank_changing.sh shape scaled to N*T samples)."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.run import make_buffer
    lanes, T = 4096, 200
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, lanes, device=device, state_mode="mixed", seed=0,
                               reward_type="distance")
    torch.manual_seed(0)
    agent = AgentResidualIntegratorModularPPO(device=device)
    agent.init(NET_DIM, env.state_dim, 1, env.n_integrator)
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    agent.init_actor_zero()
    agent.fix_K()
    buf = make_buffer(agent, env, lanes * T)

    def step():
        n = agent.explore_env(env, buf, lanes * T, 1.0, GAMMA)
        agent.update_net(buf, lanes * T, BATCH, REPEAT)
        return n
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = sum(step() for _ in range(args.steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roofline = grad_roofline(agent, lambda: agent.update_net(buf, lanes * T, BATCH, REPEAT), env.state_dim, NET_DIM,
                             "ppo_minibatch_grad = ppo_fused_dual_kernel<4, modular_actor> + ppo_grad_reduce_kernel (one minibatch of "
                             "65536 water-tank samples, D = 4)")
    out = {"metric": "env-steps/sec (rollout+update), water-tank env, 4096 parallel envs", "value": total / dt,
           "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "water tank Integrator-v2 (reward 'distance'), 4096 lanes x 200-step episodes, "
                                  "ResidualIntegratorModularPPO net_dim 128, batch 65536, repeat 8"},
           "roofline": roofline}
    emit(json_fd, out)


def bench_water_tank_256(args, device, json_fd, modular=False):
    """The reference's live water-tank script (/root/reference/run_watertank_changing.sh:20-27): ResidualPPO, net_dim 256, the
    30-float Stacking10 observation, reward 'distance'; 4096 lanes x 200-step episodes, batch 65536, repeat 8 as the other
    workloads.  The update runs on the streamed 16-tile kernels (csrc/mlp16.hip), the rollout step-wise (policy forward +
    fused residual env step per lock-step: the one-launch rollout serves widths 64 / 128).  `torch_update` is the same step with
    update_net on PyTorch-ROCm autograd + rocBLAS (use_fused_update = False): what width 256 fell back to in round 1."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO, AgentResidualPPO
    from pime_amd.elegantrl.run import make_buffer
    lanes, T = 4096, 200

    def run(fused_update, steps, warmup):
        # modular: the script's second block (run_watertank_changing.sh:11-18): ResidualIntegratorModularPPO on the Integrator
        # observation, served by the 16-tile family's modular kernels (mlp16m_forward_kernel, ppo16m_kernel) since round 3
        env = gym_control.make_vec(gym_control.WT_INTEGRATOR if modular else gym_control.WT_STACKING.format(10), lanes, device=device,
                                   state_mode="mixed", seed=0, reward_type="distance")
        torch.manual_seed(0)
        agent = (AgentResidualIntegratorModularPPO if modular else AgentResidualPPO)(device=device)
        agent.use_fused_update = fused_update
        if modular:
            agent.init(256, env.state_dim, 1, env.n_integrator)
        else:
            agent.init(256, env.state_dim, 1)
        agent.init_residual({"init_K": env.K.reshape(-1, 1)})
        agent.init_actor_zero()
        agent.fix_K()
        buf = make_buffer(agent, env, lanes * T)

        def step():
            n = agent.explore_env(env, buf, lanes * T, 1.0, GAMMA)
            agent.update_net(buf, lanes * T, BATCH, REPEAT)
            return n
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = sum(step() for _ in range(steps))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        took_hip = bool(agent._packed.get("fused"))
        roof = None
        if took_hip:
            roof = grad_roofline(agent, lambda: agent.update_net(buf, lanes * T, BATCH, REPEAT), env.state_dim, 256,
                                 ("ppo_minibatch_grad = ppo16_kernel<16, critic> + ppo16m_kernel (modular actor) + ppo_grad_reduce_kernel"
                                  if modular else
                                  "ppo_minibatch_grad = ppo16_kernel<16, critic> + ppo16_kernel<16, plain actor> + ppo_grad_reduce_kernel")
                                 + " (one minibatch of 65536 samples at width 256, streamed 16-tile family)")
        env.close()
        return total / dt, dt / steps * 1e3, took_hip, roof
    v, ms, hip, roofline = run(True, args.steps, args.warmup)
    assert hip, "width 256 did not take the HIP gradient path"
    v_t, ms_t, _, _ = run(False, max(1, args.steps // 2), 1)
    name = "Integrator env, ResidualIntegratorModularPPO" if modular else "Stacking10 env"
    out = {"metric": f"env-steps/sec (rollout+update), water-tank {name}, 4096 parallel envs, net_dim 256", "value": v,
           "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("water tank Integrator-v2 (reward 'distance'), 4096 lanes x 200-step episodes, "
                                   "ResidualIntegratorModularPPO net_dim 256 (run_watertank_changing.sh:11-18), batch 65536, repeat 8")
                      if modular else
                                  ("water tank Stacking10-v2 (30-float observation, reward 'distance'), 4096 lanes x 200-step "
                                   "episodes, ResidualPPO net_dim 256 (run_watertank_changing.sh), batch 65536, repeat 8")},
           "roofline": roofline,
           "torch_update": {"value": v_t, "unit": "env-steps/s", "ms_per_step": ms_t,
                            "what": "same step, update_net on PyTorch-ROCm autograd + rocBLAS (round 1's width-256 path)"}}
    emit(json_fd, out)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*),
    relay rank 0's stdout (the JSON line) and return non-zero if any rank fails.  The parent makes NO GPU call (importing
    torch does not initialise HIP) and never replaces itself: the ranks are ordinary child processes, each the leader of its own
    session so that its whole process group can be signalled.  No rank outlives the parent's interest in it: a failed rank, a
    SIGTERM / SIGINT to the parent, an exception in the poll loop or the overall deadline (PIME_BENCH_DEADLINE_S, default 1500 s;
    a rank stuck in RCCL init or in a collective would otherwise block the parent forever) terminate every live rank, then kill
    what is left after a grace period."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    deadline = time.monotonic() + float(os.environ.get("PIME_BENCH_DEADLINE_S", "1500"))
    procs = []

    def stop_ranks(grace=5.0):
        live = [p for p in procs if p.poll() is None]
        for sig in (signal.SIGTERM, signal.SIGKILL):
            for p in live:
                try:
                    os.killpg(p.pid, sig)   # the rank and anything it started (start_new_session: pgid == pid)
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.monotonic() + grace
            while live and time.monotonic() < t_end:
                live = [p for p in live if p.poll() is None]
                time.sleep(0.05)
            if not live:
                break

    class _Stop(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Stop(signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    chunks = []
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()   # drains rank 0's pipe for the whole run, so the rank can never block on a full one
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    log(f"rank {r} exited with code {code}; stopping the other ranks")
                    stop_ranks()   # the survivors would block in their next collective forever
            if alive and time.monotonic() > deadline:
                log(f"deadline reached with ranks {sorted(alive)} still running; stopping them")
                rc = rc or 124
                stop_ranks()
            time.sleep(0.05)
        reader.join(timeout=10)
    except _Stop as stop:
        log(f"signal {stop.args[0]} received; stopping the ranks")
        rc = 128 + int(stop.args[0])
    finally:
        stop_ranks()
        for sig, handler in old.items():
            signal.signal(sig, handler)
    out0 = b"".join(c for c in chunks if c)
    sys.stdout.buffer.write(out0)
    sys.stdout.flush()
    return rc


def launcher_selftest(json_fd):
    """--selftest-launcher: the rank plumbing alone (rendezvous, barrier, max / sum over ranks, rank 0 prints) on the gloo
    backend, no GPU -- what tests/test_bench_launcher.py runs in the GPU-less container."""
    from pime_amd import dist as pdist
    rank, world, local = pdist.env_rank_world()
    if os.environ.get("PIME_SELFTEST_HANG"):   # tests: a rank that never finishes (stuck collective), with a child of its own
        import subprocess
        subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
        open(os.environ["PIME_SELFTEST_HANG"] + f".{rank}", "w").write(str(os.getpid()))
        time.sleep(600)
    dp = pdist.init_from_env(backend="gloo", device="cpu")
    got_world = torch.distributed.get_world_size() if dp is not None else 1
    t = dp.max_over_ranks(1.0 + rank) if dp is not None else 1.0
    total = dp.sum_over_ranks(100 * (rank + 1)) if dp is not None else 100
    if dp is not None:
        dp.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps({"selftest": "launcher", "n_gpus": got_world, "max_t": t, "sum": total}) + "\n").encode())


def bench_water_tank_td3(args, device, json_fd, rank=0, world=1, dp=None):
    """BASELINE config 2 as BASELINE.json words it: water-tank Integrator env, 4096 vectorised instances, residual TD3
    (AgentResidualTD3: composed from the reference's TD3 pieces).  One step = 200 lock-steps of all lanes
    (one episode each: 819 200 transitions into the device ring) + 200 TD3 optimizer steps (one per lock-step, the reference's
    schedule counted per lock-step), batch 4096, net_dim 128; every optimizer step four hand-written launches (pime_td3_step), an
    update one HIP graph."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.run import make_buffer
    lanes, T, batch = 4096, 200, 4096
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, lanes, device=device, state_mode="mixed", seed=0, reward_type="distance")
    torch.manual_seed(0)
    agent = AgentResidualTD3(device=device)
    agent.init(NET_DIM, env.state_dim, 1)
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    agent.dp = dp   # data parallel: every rank its own lanes and ring; both gradients of a step all-reduced (ops.FusedTD3.step_dp)
    if dp is not None:
        dp.broadcast_module(agent.act, agent.cri)
        agent.act_target.load_state_dict(agent.act.state_dict()); agent.cri_target.load_state_dict(agent.cri.state_dict())
        torch.manual_seed(1000 + rank)
    buf = make_buffer(agent, env, 2 ** 21)

    def step():
        n = agent.explore_env(env, buf, lanes * T, 1.0, GAMMA)
        agent.update_net(buf, lanes * T, batch, 1)
        return n
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    t0 = time.perf_counter()
    total = sum(step() for _ in range(args.steps))
    torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_over_ranks(dt)
        total = total * world
    # roofline of the update (92 % of the step): ONE TD3 optimizer step = four hand-written launches (td3_critic_kernel,
    # td3_apply_kernel, td3_actor_kernel, td3_apply_kernel; csrc/td3_fused.hip), the 200 steps of an update replayed as one HIP
    # graph; HIP events around that replay on the stream it runs on.  Algorithmic flops per sample (net_dim 128, D = 4): actor
    # 2 * 33 664, twin critic 2 * 17 408; critic launch: target actor + target critic forwards, critic forward + backward (3x);
    # actor launch: actor forward + backward (3x), target-critic forward + input-gradient backward (2x)
    timer = KernelTimer()
    timer.enabled = True
    agent.launch_timer = timer.bracket
    agent.update_net(buf, lanes * T, batch, 1)
    timer.pairs.clear()
    agent.update_net(buf, lanes * T, batch, 1)
    torch.cuda.synchronize()
    agent.launch_timer = None
    fused = bool(getattr(agent, "_fused_td3", None))
    if fused:
        n_upd, ms_upd = timer.summary()["td3_update"]
        upd_ms = ms_upd / T
    else:   # PIME_TD3_FUSED=0: the PyTorch modules (~150 launches per step), wall clock
        t1 = time.perf_counter()
        agent.update_net(buf, lanes * T, batch, 1)
        torch.cuda.synchronize()
        upd_ms = (time.perf_counter() - t1) * 1e3 / T
    fa, fc = 2 * 33664, 2 * 17408
    flops_c, flops_a = (fa + fc + 3 * fc) * batch, (3 * fa + 2 * fc) * batch
    flops = flops_c + flops_a
    td3_tf = flops / (upd_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic_bytes(["td3_"], "r*_td3_hbm_traffic_pmc.json") if fused else (None, None)
    roofline = {"kernel": ("td3_step = td3_critic_kernel<128> + td3_apply_kernel + td3_actor_kernel<128> + td3_apply_kernel (one "
                           "optimizer step at batch 4096: gather, target nets, twin-critic and actor gradients, slab reduction, Adam, "
                           "delayed soft updates; per-kernel split in profiles/)") if fused else
                          "one TD3 optimizer step on PyTorch-ROCm (PIME_TD3_FUSED=0: ~150 autograd / rocBLAS / elementwise launches)",
                "bound": "mfma", "achieved": td3_tf, "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": td3_tf / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "traffic_source": traffic_src, "avg_launch_ms": upd_ms, "launches_per_step": 4 * T if fused else None,
                "algorithmic_flops_per_launch": flops,
                "algorithmic_flops_split": {"td3_critic_kernel": flops_c, "td3_actor_kernel": flops_a},
                "note": "avg_launch_ms is one optimizer step (four launches), timed with HIP events around the one-graph replay of an "
                        "update's 200 steps; batch 4096 = 256 workgroups of one 16-sample tile, the eight waves of a workgroup "
                        "(two per SIMD) owning one of a layer's eight output-feature tiles each (v_mfma_f32_16x16x4_f32); the exploration is ONE launch per "
                        "explore call (pime_rollout_offpolicy)"}
    out = {"metric": "env-steps/sec (rollout+update), water-tank env, 4096 parallel envs, residual TD3", "value": total / dt,
           "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "water tank Integrator-v2 (reward 'distance'), 4096 lanes/GPU x 200 lock-steps, AgentResidualTD3 "
                                  "net_dim 128, 200 optimizer steps of batch 4096 per step" +
                                  (" (one HIP graph per update)" if dp is None else
                                   " per rank (data parallel: five launches + two gradient all-reduces per step, one HIP graph per update where "
                                   "the collective can be captured)"),
                      "parallelism": f"dp{world}"},
           "roofline": roofline}
    if rank == 0:
        emit(json_fd, out)


def bench_mixed16(args, device, json_fd, rank, world, dp):
    """BASELINE config 5 (SURVEY.md section 8d cfg 5), one rank's slice: 8 192 pH lanes + 8 192 Integrator water-tank lanes in
    state_mode "mixed16" (binary16 storage of the integrated error and of the observation / reward rows; float32 / float64
    arithmetic), ensemble ranges 1.5x the registered widths (domain-randomised sweep), one ResidualIntegratorModularPPO
    net_dim 128 per env family.  One step = the two fused rollouts side by side on two HIP streams (50-step and 200-step
    episodes: 409 600 + 1 638 400 env-steps written as binary16 rows) + both PPO updates (batch 65 536, repeat 8: 50 + 200
    optimizer steps on the fused gradient kernels, each trajectory widened to float32 once per update).  Weak scaling: every rank
    owns 16 384 lanes; per optimizer step one flat-gradient all-reduce per agent."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.run import make_buffer
    from pime_amd.vec_env import VecWaterTank
    lanes = LANES // 2
    ph = gym_control.make_vec(gym_control.PH_V35, lanes, device=device, state_mode="mixed16", seed=0, env_offset=rank * lanes,
                              qww_V=(0.0045, 0.0165), qc_V=(0.00125, 0.00275))
    wt = VecWaterTank(lanes, device=device, state_mode="mixed16", seed=0, env_offset=rank * lanes, reward_type="distance",
                      a1=(0.0012, 0.0027), a2=(0.0012, 0.0027), Kp=(0.045, 0.195))
    stacks = []
    for env in (ph, wt):
        torch.manual_seed(0)
        agent = AgentResidualIntegratorModularPPO(device=device)
        agent.lambda_gae_adv = LAMBDA
        agent.init(NET_DIM, env.state_dim, 1, env.n_integrator)
        agent.init_residual({"init_K": env.K.reshape(-1, 1)})
        agent.init_actor_zero()
        agent.fix_K()
        agent.dp = dp
        if dp is not None:
            dp.broadcast_module(agent.act, agent.cri)
        buf = make_buffer(agent, env, lanes * env.max_step)
        assert buf.state.dtype == torch.float16 and agent._fused_rollout_ok(env)
        stacks.append((env, agent, buf, torch.cuda.Stream(device=device)))
    torch.manual_seed(1000 + rank)

    def step():
        cur = torch.cuda.current_stream()
        total = 0
        for env, agent, buf, stream in stacks:     # the two halves of the batch roll out side by side
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                total += agent.explore_env(env, buf, lanes * env.max_step, 1.0, GAMMA)
        for env, agent, buf, stream in stacks:
            cur.wait_stream(stream)
        for env, agent, buf, stream in stacks:
            agent.update_net(buf, lanes * env.max_step, BATCH, REPEAT)
        return total

    def sync():
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    total = sum(step() for _ in range(args.steps))
    sync()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_over_ranks(dt)
        total = dp.sum_over_ranks(total)
    roofline = None
    if dp is None:   # (the event bracket forces the two-graph step sequence: single-rank runs only)
        env_w, agent_w, buf_w, _ = stacks[1]
        roofline = grad_roofline(agent_w, lambda: agent_w.update_net(buf_w, lanes * env_w.max_step, BATCH, REPEAT), env_w.state_dim,
                                 NET_DIM, "ppo_minibatch_grad = ppo_fused_dual_kernel<4, modular_actor> + ppo_grad_reduce_kernel (one "
                                 "minibatch of 65536 samples of the water-tank half, widened from the binary16 trajectory; 200 of the "
                                 "step's 250 optimizer steps)")
    if rank == 0:
        out = {"metric": "env-steps/sec (rollout+update), mixed pH + water-tank batch, fp16 state, 16384 lanes/GPU",
               "value": total / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE config 5 rank slice: 8192 pH v35 lanes x 50-step episodes + 8192 water-tank "
                                      "Integrator-v2 lanes (reward 'distance') x 200-step episodes, state_mode mixed16 (binary16 "
                                      "I / observation / reward storage, f32 math, f64 x), ensemble ranges x1.5, "
                                      "ResidualIntegratorModularPPO net_dim 128 per family, batch 65536, repeat 8",
                          "lanes_per_gpu": LANES, "parallelism": f"dp{world}"},
               "roofline": roofline}
        emit(json_fd, out)
    if dp is not None:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="ph", choices=["ph", "wt", "wt_td3", "wt256", "wtmod256", "mixed16"],
                    help="ph: the headline config (BASELINE config 3); wt: config 2, water tank, 4096 lanes x 200 steps "
                         "(reported for DESIGN.md; the headline metric is the ph line)")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # no launcher: become one, before anything touches the GPU
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout must carry exactly ONE JSON line: libraries that print banners to fd 1 (RCCL prints its version block at
    # communicator creation) are sent to stderr for the whole run, and the JSON is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from pime_amd import dist as pdist
    rank, world, local = pdist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.selftest_launcher:
        return launcher_selftest(json_fd)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # PIME_BENCH_REHEARSE=1: every rank on cuda:0 with gloo collectives -- a rehearsal of the launcher and the data-parallel
    # step sequence on a one-GPU box (RCCL refuses two ranks on one device); its numbers mean nothing
    rehearse = os.environ.get("PIME_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local
    device = f"cuda:{dev_index}"
    torch.cuda.set_device(dev_index)
    dp = (pdist.init_from_env(backend="gloo" if rehearse else "nccl", device=device)
          if (world > 1 or os.environ.get("PIME_FORCE_DP") == "1") else None)
    if dp is not None:
        world = torch.distributed.get_world_size()   # the RCCL communicator's size is what the JSON line reports

    if args.workload == "mixed16":
        return bench_mixed16(args, device, json_fd, rank, world, dp)
    if args.workload == "wt":
        return bench_water_tank(args, device, json_fd)
    if args.workload == "wt256":
        return bench_water_tank_256(args, device, json_fd)
    if args.workload == "wtmod256":
        return bench_water_tank_256(args, device, json_fd, modular=True)
    if args.workload == "wt_td3":
        return bench_water_tank_td3(args, device, json_fd, rank, world, dp)
    env, agent, buf = build_stack(device, rank, world, dp)
    timer = KernelTimer()
    # time the hand-written kernels where the agent calls them
    import pime_amd.ops as ops
    agent.state_value = timer.wrap("mlp_forward<critic> value pass", agent.state_value)
    agent.policy_mean = timer.wrap("mlp_forward<modular_actor> rollout", agent.policy_mean)
    env.step_residual = timer.wrap("ph_step_kernel (fused residual)", env.step_residual)
    env.rollout = timer.wrap("rollout_ph_kernel (50 steps, one launch)", env.rollout)
    agent.backend.gae = timer.wrap("gae_scan_kernel", agent.backend.gae)
    # "ppo_minibatch_grad" (the gradient launches of one minibatch) is timed with HIP events in ONE extra, untimed update
    # after the measured region: bracketing it forces the two-graph step sequence, while the measured steps replay one
    # graph per optimizer step.

    def sync():
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
            torch.cuda.synchronize()

    log("stack built; warmup")
    for i in range(args.warmup):
        one_step(env, agent, buf)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    sync()
    timer.enabled = True
    t_roll = t_upd = 0.0
    t0 = time.perf_counter()
    total = 0
    for _ in range(args.steps):
        total += one_step(env, agent, buf)
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    log(f"timed region: {dt:.3f}s for {args.steps} steps")
    if dp is not None:
        dt = dp.max_over_ranks(dt)
        total = dp.sum_over_ranks(total)

    # rollout / update split (one extra untimed step, synchronised between the halves)
    sync()
    a = time.perf_counter()
    agent.explore_env(env, buf, LANES * T_EP, 1.0, GAMMA)
    torch.cuda.synchronize()
    b = time.perf_counter()
    agent.update_net(buf, LANES * T_EP, BATCH, REPEAT)
    torch.cuda.synchronize()
    c = time.perf_counter()
    t_roll, t_upd = b - a, c - b
    # per-launch duration of the gradient launches: one more update with HIP events around every minibatch's launches
    grad_timer = KernelTimer()
    grad_timer.enabled = True
    agent.launch_timer = grad_timer.bracket
    agent.update_net(buf, LANES * T_EP, BATCH, REPEAT)   # first use of the two-graph sequence: captures it
    grad_timer.pairs.clear()
    agent.update_net(buf, LANES * T_EP, BATCH, REPEAT)
    torch.cuda.synchronize()
    agent.launch_timer = None

    if rank != 0:
        if dp is not None:
            torch.distributed.destroy_process_group()
        return
    ks = timer.summary()
    per_step = {k: n * ms / args.steps for k, (n, ms) in ks.items()}
    n_g, ms_g = grad_timer.summary()["ppo_minibatch_grad"]
    ks["ppo_minibatch_grad"] = (n_g * args.steps, ms_g)       # n_g launches per update, one update per step
    per_step["ppo_minibatch_grad"] = n_g * ms_g
    dominant = max(per_step, key=per_step.get)
    n_dom, ms_dom = ks[dominant]
    if dominant == "ppo_minibatch_grad":
        achieved = GRAD_FLOPS_PER_SAMPLE * BATCH / (ms_dom * 1e-3) / 1e12
        # the gradient kernels of the profile: ppo_fused_dual_kernel (round 3: actor and critic bodies in one grid) or the two
        # ppo_fused_kernel launches of rounds 1-2, plus the slab reduction
        traffic, traffic_src = pmc_traffic_bytes(["ppo_fused_", "ppo_grad_reduce_kernel"])
        roofline = {"kernel": "ppo_minibatch_grad = ppo_fused_dual_kernel<4, modular_actor> (actor and critic bodies in one grid) + "
                              "ppo_grad_reduce_kernel (one minibatch of 65536: forward, loss, backward and weight "
                              "gradients of both nets, slab reduction; per-kernel split in profiles/)", "bound": "mfma",
                    "achieved": achieved,
                    "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                    "traffic": traffic,
                    "traffic_source": f"{traffic_src} (rocprofv3 --pmc of this bench, separate FETCH_SIZE / WRITE_SIZE passes, "
                                      "FETCH_SIZE x2)" if traffic_src else None,
                    "launches_per_step": n_dom / args.steps, "avg_launch_ms": ms_dom,
                    "algorithmic_flops_per_launch": GRAD_FLOPS_PER_SAMPLE * BATCH,
                    "clock_note": "peak is the guide's figure at 2.4 GHz; s_memtime / s_memrealtime inside these kernels read 2.17 GHz with "
                                  "all 256 compute units busy (profiles/r04_v_clock_and_group_timing.txt: a recorded measurement, not "
                                  "taken in this run), at which the f32 matrix peak is 142 TFLOP/s"}
        if os.environ.get("PIME_MLP16"):   # the 16-tile family serves this width: other kernels, no PMC file of theirs
            roofline["kernel"] = ("ppo_minibatch_grad = ppo16m_kernel<8> (modular actor) + ppo16_kernel<8> (critic) + ppo_grad_reduce_kernel "
                                  "(PIME_MLP16=1: the streamed 16-tile family at width 128)")
            roofline["traffic"], roofline["traffic_source"] = None, None
        share = bf16x3_share(NET_DIM)
        if share > 0:
            peak = 1.0 / (share / (BF16_MFMA_PEAK_TFLOPS / 6.0) + (1.0 - share) / F32_MFMA_PEAK_TFLOPS)
            roofline.update({"peak": peak, "frac": achieved / peak, "frac_of_f32_peak": achieved / F32_MFMA_PEAK_TFLOPS,
                             "peak_note": f"bf16x3: {share:.2f} of the flops as six v_mfma_f32_16x16x32_bf16 per product block "
                                          f"({BF16_MFMA_PEAK_TFLOPS / 6:.1f} TFLOP/s f32-equivalent), the rest f32 MFMA; achieved counts "
                                          "f32-equivalent flops"})
    elif "critic" in dominant or "actor" in dominant:
        kind = "critic" if "critic" in dominant else "modular_actor"
        rows = LANES * T_EP if kind == "critic" else LANES
        achieved = MLP_FLOPS_PER_ROW[kind] * rows / (ms_dom * 1e-3) / 1e12
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                    "launches_per_step": n_dom / args.steps, "avg_launch_ms": ms_dom}
    else:
        achieved = PH_STEP_BYTES * LANES / (ms_dom * 1e-3) / 1e9
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches_per_step": n_dom / args.steps,
                    "avg_launch_ms": ms_dom}
    n_v, ms_v = ks["mlp_forward<critic> value pass"]
    v_tf = MLP_FLOPS_PER_ROW["critic"] * LANES * T_EP / (ms_v * 1e-3) / 1e12
    ro_key = "rollout_ph_kernel (50 steps, one launch)"
    if ro_key in ks:   # fused rollout: policy forward + noise + env step + buffer writes for a whole episode per launch
        n_env, ms_env = ks[ro_key]
        ro_tf = MLP_FLOPS_PER_ROW["modular_actor"] * LANES * T_EP / (ms_env * 1e-3) / 1e12
        roofline_env = {"kernel": "rollout_kernel<4, modular_actor, pH, 16-lane tiles> (16384 lanes x 50 steps per launch)", "bound": "mfma",
                        "achieved": ro_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ro_tf / F32_MFMA_PEAK_TFLOPS,
                        "traffic": None, "avg_launch_ms": ms_env,
                        "note": "one 16-lane tile per wave, one wave on each of the 1 024 SIMDs, a serial chain of 512 "
                                "v_mfma_f32_16x16x4_f32 (6.8 us) per env step plus the env arithmetic: latency bound by design "
                                "(round 3: 32-lane tiles left half of the SIMDs idle, 1.20 ms -> 0.69 ms per launch); the "
                                "step-per-launch env kernels move 5.4 TB/s (pH) / 4.6 TB/s (WT) of PMC-counted HBM traffic at 4M "
                                "lanes, profiles/r03_e_env_pmc.json"}
    else:
        n_env, ms_env = ks["ph_step_kernel (fused residual)"]
        env_gbs = PH_STEP_BYTES * LANES / (ms_env * 1e-3) / 1e9
        roofline_env = {"kernel": "ph_step_kernel (fused residual)", "bound": "hbm", "achieved": env_gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": env_gbs / HBM_PEAK_GBS,
                        "traffic": pmc_traffic_bytes(["ph_step_kernel"])[0], "algorithmic_bytes_per_launch": PH_STEP_BYTES * LANES,
                        "avg_launch_ms": ms_env, "note": "16384-lane launch moves 1.5 MB: launch-latency bound"}
    out = {
        "metric": "env-steps/sec (rollout+update), pH env, 16384 parallel envs",
        "value": total / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "pH v35, 16384 lanes/GPU x 50-step episodes, ensemble resampled every episode, "
                               "ResidualIntegratorModularPPO net_dim 128, batch 65536, repeat 8 (100 optimizer steps)",
                   "lanes_per_gpu": LANES, "episode_len": T_EP, "batch": BATCH, "repeat_times": REPEAT,
                   "state_mode": "mixed (f32 state, f64 x/A/B/C)", "parallelism": f"dp{world}"},
        "roofline": roofline,
        "roofline_env": roofline_env,
        "roofline_value_pass": {"kernel": "mlp_forward_kernel<4, critic> (819200 rows)", "bound": "mfma", "achieved": v_tf,
                                "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": v_tf / F32_MFMA_PEAK_TFLOPS,
                                "avg_launch_ms": ms_v},
        "breakdown_ms": {"rollout": t_roll * 1e3, "update": t_upd * 1e3,
                         "hand_written_kernels_per_step": per_step},
    }
    if not args.no_cpu_baseline and world == 1:
        log("cpu baseline (oracle env + torch CPU update) ...")
        out["cpu_baseline"] = cpu_baseline()
        log("cpu baseline done")
    emit(json_fd, out)
    if dp is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
