#!/usr/bin/env python3
"""bench.py -- headline benchmark: input Msamples/s through FIR-decimate + FM demod.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one 20 MS/s stream of 2^30 synthetic int16 IQ samples
resident in HBM, 64-tap Hann low-pass, decimate by 10, quadrature discriminator -> float32.
A "step" is one pass of the fused HIP kernel over the whole 2^30-sample block.

N > 1 (launched by torch.distributed.run, one rank per GPU): the stream is N * 2^30 samples and
rank r owns time block r with its FIR history halo resident -- no collective in the
timed loop, scaling "weak".  The final RCCL gather of the decimated outputs to rank 0 runs once
after the timed steps and is reported separately (gather_ms).

The JSON line carries `roofline` (HIP-event time of the kernel itself, measured on the launch
stream, against the 8 TB/s HBM peak with 4.4 algorithmic bytes per input sample) and, at N = 1,
`cpu_baseline` (the reference's own chain from oracle/_ref timed on this box's host, one thread).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--log2n", type=int, default=30, help="samples per GPU = 2^log2n")
    ap.add_argument("--taps", type=int, default=64)
    ap.add_argument("--decim", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--outputs-per-lane", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--config", choices=["stream", "channels"], default="stream",
                    help="stream (default): BASELINE configs[1], one long stream, time-block shards, weak scaling; "
                         "channels: BASELINE configs[2], --channels independent channels of 2^--chan-log2n samples "
                         "block-partitioned over the ranks (strong scaling), one batched launch per rank")
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--chan-log2n", type=int, default=20)
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="upper bound of the clock-stabilising pre-warm that runs before --warmup (0 = off)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other BASELINE configs (channelizer, waterfall, APT chain) that the N = 1 stream run "
                         "measures after the headline and reports under \"secondary\"")
    ap.add_argument("--secondary-cpu-seconds", type=float, default=3.0, help="budget of each secondary config's CPU leg (0 = skip)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: every rank uses device 0 and the collectives go "
                         "over gloo on host tensors (checks the sharded path; the numbers mean nothing)")
    return ap.parse_args()


def cpu_baseline(O, taps, D, budget_s):
    """reference chain (scale -> polyphase_decimate -> polar_discriminator), one host thread, on a
    bounded sample of the same synthetic signal"""
    n = 1 << 24
    iq = O.synth_fm_i16(n, fs=20.0e6, ftone=1000.0, dev=5000.0, amp=2000.0)
    scale = 1.0 / (2.0 * np.pi)
    if O.have_ref():
        kind = "reference"
        t1 = O.ref_time_chain_i16(iq, D, taps, O.MODE_FM, scale, 1)
        reps = max(1, int(budget_s / max(t1, 1e-3)) - 1)
        t = O.ref_time_chain_i16(iq, D, taps, O.MODE_FM, scale, reps)
    else:
        kind = "port"
        t0 = time.perf_counter()
        O.chain_i16(iq, D, taps, O.MODE_FM, scale)
        t1 = time.perf_counter() - t0
        reps = max(1, int(budget_s / max(t1, 1e-3)) - 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            O.chain_i16(iq, D, taps, O.MODE_FM, scale)
        t = time.perf_counter() - t0
    return {"value": round(n * reps / t / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": f"{reps} passes over the first 2^24 samples of the same synthetic FM stream, "
                      f"{t:.1f} s, float64 as the reference, host nproc={os.cpu_count()}"}


def cpu_all_cores(O, taps, D, budget_s):
    """the oracle's C port of the same chain on every host core of the box's share (SURVEY 8d ii):
    one independent 2^22-sample block per thread (ctypes calls release the GIL)"""
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, os.cpu_count() or 1))
    n = 1 << 22
    iq = O.synth_fm_i16(n, fs=20.0e6, ftone=1000.0, dev=5000.0, amp=2000.0)
    blocks = [iq.copy() for _ in range(threads)]
    scale = 1.0 / (2.0 * np.pi)
    t0 = time.perf_counter()
    O.chain_i16(iq, D, taps, O.MODE_FM, scale)
    reps = max(1, int(budget_s / max(time.perf_counter() - t0, 1e-3)))

    def work(b):
        for _ in range(reps):
            O.chain_i16(b, D, taps, O.MODE_FM, scale)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, blocks))
    t = time.perf_counter() - t0
    return {"value": round(n * reps * threads / t / 1e6, 2), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads x {reps} passes over a 2^22-sample block each, {t:.1f} s"}


def launch_ranks(a):
    """`python bench.py --gpus N` on its own: start N ranks (one process per GPU) through
    torch.distributed.run as a child process and relay its output.  This process has not touched the
    GPU (torch is not imported yet), and it starts a child rather than replacing itself."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def secondary(cpu_seconds):
    """The other BASELINE configs on this GPU, once each, warm: kernel time (HIP events), roofline, spot
    parity against the oracle and a bounded CPU baseline (tools/bench_configs.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as BC
    out = {}
    import bench_channelizer as BCH

    def nco(_iters, _cpu):
        return BCH.run("cfg3 with one NCO per channel: 1024 separate 2^20-sample streams, per-channel mixer -> 64 taps /10 -> FM "
                       "(sdr_fir_execute_channels_dev; SURVEY 8 a7)", 1 << 20, False, emit=False)

    def nco_shared(_iters, _cpu):
        return BCH.run("channelizer proper: 1024 NCOs on ONE shared 2^22-sample wideband stream", 1 << 22, True, emit=False)

    for name, fn in (("cfg3_channelizer", BC.cfg3), ("cfg3_channelizer_per_channel_nco", nco), ("channelizer_shared_wideband_input", nco_shared),
                     ("cfg4_waterfall", BC.cfg4), ("cfg5_apt_chain", BC.cfg5)):
        t0 = time.perf_counter()
        try:
            r = fn(20, cpu_seconds)
            r["wall_s"] = round(time.perf_counter() - t0, 1)
            out[name] = r
        except Exception as e:                      # the headline stands; say what failed
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import torch
    import sdr_amd
    from oracle import oracle as O

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.rehearse:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    assert world == a.gpus, f"--gpus {a.gpus} but the process group has {world} rank(s)"
    dev = torch.device("cuda", local)
    sdr_amd.init(local)

    D, T = a.decim, a.taps
    taps = sdr_amd.create_fir_filter(T, 1.0 / D, "hann")     # cutoff at the decimated Nyquist
    plan = sdr_amd.FirPlan(taps, D, mode=sdr_amd.MODE_FM, scale=1.0 / (2.0 * np.pi),
                           outputs_per_lane=a.outputs_per_lane, threads=a.threads,
                           pipeline=a.pipeline)
    from sdr_amd import shard
    st = torch.cuda.current_stream().cuda_stream
    channels = a.config == "channels"
    if channels:
        # BASELINE configs[2]: independent channels, channel-major; rank r owns a contiguous block of
        # channels (shard.plan_channel_shards) and filters them in ONE batched launch
        n_chan = 1 << a.chan_log2n
        c_first, c_mine = shard.plan_channel_shards(a.channels, world)[rank]
        cnt_ch = plan.out_count(n_chan)
        n_blk = c_mine * n_chan                    # samples this rank processes per step
        n_all = a.channels * n_chan                # samples the whole job processes per step
        d_in = torch.empty(2 * max(n_blk, 1) + 8, dtype=torch.int16, device=dev)
        d_out = torch.empty(max(c_mine * cnt_ch, 1) + 64, dtype=torch.float32, device=dev)
        for c in range(c_mine):                    # per-channel tone 300..3000 Hz, seed = base + channel
            gc = c_first + c
            sdr_amd.capi.synth_fm_i16_dev(d_in[2 * c * n_chan:], n_chan, fs=1.0e6,
                                          ftone=300.0 + 2700.0 * gc / max(a.channels - 1, 1), dev=5000.0,
                                          noise_lsb=2, seed=100 + gc, stream=st)
        torch.cuda.synchronize()

        def step():
            if c_mine:
                plan.execute_batched_dev(d_in, n_chan, n_chan, c_mine, d_out, cnt_ch, stream=st)
    else:
        n_blk = 1 << a.log2n                       # samples per GPU
        n_total = n_blk * world
        n_all = n_total
        shards = shard.plan_time_shards(n_total, D, T, shard.MODE_FM, world)
        me = shards[rank]                          # this rank's slice of the final output
        cnt_total = plan.out_count(n_total)
        assert shards[-1].out_end == cnt_total
        ob, oe = me.out_begin, me.out_end
        lo, hi = me.lo, me.hi                      # resident window = own block + FIR history halo
        n_res = hi - lo

        d_in = torch.empty(2 * n_res + 8, dtype=torch.int16, device=dev)
        d_out = torch.empty(oe - ob + (T + 2 * D) // D + 64, dtype=torch.float32, device=dev)
        sdr_amd.capi.synth_fm_i16_dev(d_in, n_res, n0=lo, fs=20.0e6, ftone=1000.0, dev=5000.0,
                                      amp=2000.0, noise_lsb=0, seed=1, stream=st)
        torch.cuda.synchronize()

        def step():
            plan.execute_range_dev(d_in, lo, n_res, n_total, ob, oe, d_out, stream=st)

    # Clock-stabilising pre-warm, independent of --warmup: the first launches after idle run 10-15 %
    # slow (clock ramp), so batches of launches are event-timed until two consecutive batches agree
    # within 1 %, bounded by --prewarm-seconds.  Untimed: nothing of it enters `value`.
    prewarm = {"launches": 0, "batches": 0, "ms_last_batch": None, "stable": False, "wall_s": 0.0}
    if a.prewarm_seconds > 0:
        pw0 = time.perf_counter()
        prev = None
        batch = 20
        while time.perf_counter() - pw0 < a.prewarm_seconds:
            b0 = torch.cuda.Event(enable_timing=True)
            b1 = torch.cuda.Event(enable_timing=True)
            b0.record()
            for _ in range(batch):
                step()
            b1.record()
            b1.synchronize()
            cur = b0.elapsed_time(b1) / batch
            prewarm["launches"] += batch
            prewarm["batches"] += 1
            prewarm["ms_last_batch"] = round(cur, 4)
            if prev is not None and prewarm["batches"] >= 3 and abs(cur - prev) <= 0.01 * cur:
                prewarm["stable"] = True
                break
            prev = cur
        prewarm["wall_s"] = round(time.perf_counter() - pw0, 3)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream (st IS torch's current stream) bracket the same K launches
    # (one event per step as well: min / median over the timed steps)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    ev0, ev1 = evs[0], evs[-1]
    t0 = time.perf_counter()
    ev0.record()
    for i in range(a.steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- kernel time (roofline): the K launches of the timed region, back to back on one stream,
    # between the two HIP events; each launch processes this rank's n_blk samples
    ms_kernel = ev0.elapsed_time(ev1) / a.steps
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
    ms_min, ms_med = per_step[0], per_step[len(per_step) // 2]
    n_kernel = n_blk
    bytes_per_sample = 4.0 + 4.0 / D
    achieved = bytes_per_sample * n_kernel / (ms_kernel * 1e-3) / 1e9

    # ---- spot parity at full size: three slices (start, past 2^31 bytes, end) vs the oracle --
    parity = None
    if rank == 0 and channels:
        worst = 0.0
        for c in sorted({0, c_mine // 2, c_mine - 1}) if c_mine else []:
            blk = d_in[2 * c * n_chan: 2 * (c + 1) * n_chan].cpu().numpy()
            want = O.chain_i16(blk, D, taps, O.MODE_FM, 1.0 / (2.0 * np.pi))
            got = d_out[c * cnt_ch: (c + 1) * cnt_ch].cpu().numpy().astype(np.float64)
            worst = max(worst, float(np.abs(got - want).max()))
        parity = worst
        assert worst <= 1e-5 * 0.5, f"full-size parity failed: {worst}"
    elif rank == 0:
        worst = 0.0
        out_n = oe - ob
        for j0 in (0, out_n // 2 + 12345, out_n - 4096):
            j0 = max(0, min(j0, out_n - 4096))
            J = ob + j0                           # first output of the slice (stream index)
            s_lo = max(lo, J * D - (T - 1))
            s_hi = min(hi, (J + 4096) * D + 1)
            blk = d_in[2 * (s_lo - lo): 2 * (s_hi - lo)].cpu().numpy()
            y = O.fir_decimate(O.unpack_scale_i16(blk), D, taps, pre=J * D - s_lo)
            want = O.polar_discriminator(y, 1.0 / (2.0 * np.pi))[:4096]
            got = d_out[j0: j0 + want.size].cpu().numpy().astype(np.float64)
            worst = max(worst, float(np.abs(got - want).max()))
        parity = worst
        assert worst <= 1e-5 * 0.5, f"full-size parity failed: {worst}"

    gather_ms = gather_error = None
    if dist and not a.no_gather:
        # final collection of the decimated output on rank 0 over xGMI (RCCL gather), once per job
        counts = [k * cnt_ch for (_c0, k) in shard.plan_channel_shards(a.channels, world)] if channels \
            else [sh.n_out for sh in shards]
        cnt_total = sum(counts)
        torch.cuda.synchronize(); dist.barrier()
        g0 = time.perf_counter()
        try:
            gathered = shard.gather_outputs(d_out.cpu() if a.rehearse else d_out, counts, dist, dst=0)
            torch.cuda.synchronize(); dist.barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
        except Exception as e:                    # the timed result above stands; report, do not lose the line
            gathered, gather_error = None, f"{type(e).__name__}: {e}"
            print(f"rank {rank}: final gather failed: {gather_error}", file=sys.stderr)
        if rank == 0 and gathered is not None:
            assert gathered.numel() == cnt_total
            if a.rehearse and channels:
                # the gathered channel blocks must equal all channels filtered in one launch, bit for bit
                d_all = torch.empty(2 * n_all + 8, dtype=torch.int16, device=dev)
                for gc in range(a.channels):
                    sdr_amd.capi.synth_fm_i16_dev(d_all[2 * gc * n_chan:], n_chan, fs=1.0e6,
                                                  ftone=300.0 + 2700.0 * gc / max(a.channels - 1, 1), dev=5000.0,
                                                  noise_lsb=2, seed=100 + gc, stream=st)
                d_one = torch.empty(cnt_total + 8, dtype=torch.float32, device=dev)
                plan.execute_batched_dev(d_all, n_chan, n_chan, a.channels, d_one, cnt_ch, stream=st)
                torch.cuda.synchronize()
                n_bad = int((gathered != d_one[:cnt_total].cpu()).sum())
                assert n_bad == 0, n_bad
                print(f"rehearsal: gathered channel blocks vs one launch over all {a.channels} channels: "
                      f"{cnt_total} outputs, {n_bad} differ", file=sys.stderr)
            elif a.rehearse:
                # the gathered shards must equal the whole stream filtered in one piece, bit for bit
                d_all = torch.empty(2 * n_total + 8, dtype=torch.int16, device=dev)
                sdr_amd.capi.synth_fm_i16_dev(d_all, n_total, n0=0, fs=20.0e6, ftone=1000.0, dev=5000.0,
                                              amp=2000.0, noise_lsb=0, seed=1, stream=st)
                d_one = torch.empty(cnt_total + 8, dtype=torch.float32, device=dev)
                plan.execute_dev(d_all, n_total, d_one, cnt_total, stream=st)
                torch.cuda.synchronize()
                one = d_one[:cnt_total].cpu()
                # every output has one summation order whatever the tiling, so the shards must
                # reproduce the one-piece result bit for bit
                err = float((gathered - one).abs().max())
                n_bad = int((gathered != one).sum())
                assert n_bad == 0, (err, n_bad)
                print(f"rehearsal: gathered shards vs one-piece result: {cnt_total} outputs, max |diff| {err:.2e}, "
                      f"{n_bad} differ in the last bit", file=sys.stderr)

    # the same launch once more in the library's own event loop, after everything is warm (last:
    # it treats the resident window as a stream of its own and overwrites d_out)
    if channels:
        ms_warm = plan.time_dev(d_in, n_chan, d_out, max(5, a.steps), in_stride=n_chan, n_channels=max(c_mine, 1),
                                out_stride=cnt_ch, stream=st)
    else:
        ms_warm = plan.time_dev(d_in, n_res if world > 1 else n_blk, d_out, max(5, a.steps), stream=st)

    if rank == 0:
        value = n_all * a.steps / dt / 1e6
        info = plan.info()
        # the arithmetic the kernel that ran computes in (outputs are float32 either way)
        arith = {"mfma_i8": "int8-split exact integer (int8 x int8 -> int32 matrix products of the samples' bytes and "
                            "31-bit fixed-point taps), one float32 combination per output; float32 out",
                 }.get(info["kernel"], "f32 (float32 taps and accumulation); float32 out")
        info["arithmetic"] = arith
        # HBM bytes per launch from the PMC counters cannot be collected inside this process (they
        # need rocprofv3 around it): the figure comes from the tracked summary of the last counter run
        # of this same command and is labelled with its source
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("fir_bytes_per_launch")
                traffic_source = "profiles/hbm_traffic.json, counter run of commit %s, profile tag %s (%s)" % (
                    tj.get("commit", "?"), tj.get("profile_tag", "?"),
                    tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, not this run"))
            except Exception:
                traffic = None
        line = {
            "metric": "input Msamples/s through FIR-decimate+FM demod",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if channels else "weak", "vs_baseline": None,
            "dtype": arith, "data": "synthetic", "ranks_in_group": dist.get_world_size() if dist else 1,
            "config": ({"workload": f"FM channelizer, {a.channels} independent channels x 2^{a.chan_log2n} int16 IQ samples, "
                                    f"{T}-tap Hann LPF, decimate by {D}, discriminator (BASELINE configs[2]); "
                                    f"channels block-partitioned over the ranks, one batched launch per rank",
                        "channels": a.channels, "samples_per_channel": n_chan, "channels_rank0": c_mine,
                        "taps": T, "decim": D, "sharding": "channel blocks" if world > 1 else "none", "kernel": info}
                       if channels else
                       {"workload": f"FM chain, single 20 MSps stream, 2^{a.log2n}-sample int16 IQ block per GPU, "
                                    f"{T}-tap Hann LPF, decimate by {D}, discriminator (BASELINE configs[1])",
                        "samples_per_gpu": n_blk, "taps": T, "decim": D,
                        "sharding": "time blocks with FIR history halo" if world > 1 else "none",
                        "kernel": info}),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(ms_kernel, 4),
                         "kernel_ms_min": round(ms_min, 4), "kernel_ms_median": round(ms_med, 4),
                         "kernel_ms_after_timed_region": round(ms_warm, 4),
                         "algorithmic_bytes_per_sample": bytes_per_sample},
            "parity_max_abs_err": parity,
            "prewarm": prewarm,
        }
        if gather_ms is not None:
            line["gather_ms"] = round(gather_ms, 3)
        if gather_error is not None:
            line["gather_error"] = gather_error
        if world == 1 and a.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(O, taps, D, a.cpu_seconds)
            line["cpu_baseline_all_cores"] = cpu_all_cores(O, taps, D, a.cpu_seconds / 3.0)
        if world == 1 and not channels and not a.no_secondary:
            del d_in, d_out
            torch.cuda.empty_cache()
            line["secondary"] = secondary(a.secondary_cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
