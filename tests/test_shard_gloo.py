"""CPU tests of the N>1 path: the sharding plans and the final gather, world_size 2 over gloo.
The per-rank compute is stood in for by the CPU oracle (there is no GPU here); what is under
test is the plan (ranges + halos) and the collective.  On GPUs bench.py runs the same plan with
sdr_fir_execute_range_dev and backend nccl (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdr_amd import shard  # noqa: E402


def test_time_shard_plan_covers_outputs_and_halos():
    for (n, D, T, mode, world) in [(1_000_003, 10, 64, shard.MODE_FM, 2), (1 << 20, 10, 511, shard.MODE_ENV, 8),
                                   (777, 15, 195, shard.MODE_FM, 4), (50, 10, 64, shard.MODE_FM, 8),
                                   (1 << 30, 10, 64, shard.MODE_FM, 8)]:
        sh = shard.plan_time_shards(n, D, T, mode, world)
        cnt = shard.out_count(n, D, mode)
        assert sh[0].out_begin == 0 and sh[-1].out_end == cnt
        for a, b in zip(sh, sh[1:]):
            assert a.out_end == b.out_begin
        for s in sh:
            if s.n_out == 0:
                continue
            assert 0 <= s.lo <= max(0, s.out_begin * D - (T - 1))
            need_hi = (s.out_end - 1 + (1 if mode == shard.MODE_FM else 0)) * D + 1
            assert s.hi >= min(n, need_hi) and s.hi <= n
    # the ArrayFire loops' count rule (one more y unless D divides N): the last shard reaches it
    for (n, D, T, mode, world) in [(131072, 5, 101, shard.MODE_FM, 2), (100_001, 10, 64, shard.MODE_ENV, 3), (7, 10, 64, shard.MODE_IQ, 2)]:
        sh = shard.plan_time_shards(n, D, T, mode, world, shard.COUNT_AF_SEQ)
        ny = (n - 1) // D + 1
        assert sh[-1].out_end == shard.out_count(n, D, mode, shard.COUNT_AF_SEQ) == (ny - 1 if mode == shard.MODE_FM else ny)
        last = [s for s in sh if s.n_out][-1]
        assert last.hi == min(n, (ny - 1) * D + 1)
    assert shard.plan_channel_shards(1024, 8) == [(128 * r, 128) for r in range(8)]
    assert shard.plan_channel_shards(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    fs = shard.plan_frame_shards(1 << 20, 4096, 2048, 4)
    assert fs[0][0] == 0 and fs[-1][1] == ((1 << 20) - 4096) // 2048 + 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    n, D, T = 200_007, 10, 64
    iq = O.synth_fm_i16(n, fs=1.0e6, ftone=1000.0, dev=5000.0)        # every rank can regenerate the stream
    h = O.create_fir_filter(T, 0.1, "hann")
    sh = shard.plan_time_shards(n, D, T, mode, world)[rank]
    # this rank only touches its resident window [lo, hi)
    blk = iq[2 * sh.lo: 2 * sh.hi]
    x = O.unpack_scale_i16(blk)
    # the oracle emits floor(n/D) outputs; the successor y the discriminator needs sits exactly on
    # the last resident sample, so give the oracle D-1 more (zero) samples to count it
    x = np.concatenate([x, np.zeros(D - 1, np.complex128)])
    pre = sh.out_begin * D - sh.lo
    y = O.fir_decimate(x, D, h, pre=pre)
    if mode == shard.MODE_FM:
        local = O.polar_discriminator(y, 1 / (2 * np.pi))[: sh.n_out]
    else:
        local = O.envelope(y)[: sh.n_out]
    assert local.size == sh.n_out
    counts = [s.n_out for s in shard.plan_time_shards(n, D, T, mode, world)]
    got = shard.gather_outputs(torch.from_numpy(local.copy()), counts, dist, dst=0)
    if rank == 0:
        whole = O.chain_i16(iq, D, h, O.MODE_FM if mode == shard.MODE_FM else O.MODE_ENV)
        q.put(float(np.abs(got.numpy() - whole).max()) if got.numel() == whole.size else -1.0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", [shard.MODE_FM, shard.MODE_ENV])
def test_time_shards_gather_equals_whole_gloo(mode):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    err = q.get()
    assert 0 <= err <= 1e-12, err


def _chan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    n_ch, n, D = 6, 20_000, 10
    h = O.create_fir_filter(64, 0.1, "hann")
    c0, k = shard.plan_channel_shards(n_ch, world)[rank]
    outs = []
    for c in range(c0, c0 + k):
        iq = O.synth_fm_i16(n, fs=1.0e6, ftone=300.0 + 100.0 * c, dev=5000.0)
        outs.append(O.chain_i16(iq, D, h, O.MODE_FM))
    # the channelizer proper: every rank holds the same wideband buffer and runs ITS channels' mixers
    # (sdr_fir_execute_channels_dev with chan_first = c0 on a GPU; here the oracle per channel)
    wide = O.unpack_scale_i16(O.synth_fm_i16(n, fs=1.0e6, ftone=700.0, dev=20000.0))
    frs = [0.05 * c - 0.12 for c in range(n_ch)]
    for c in range(c0, c0 + k):
        y = O.fir_decimate(O.frequency_shift(wide, frs[c]), D, h)
        outs[c - c0] = outs[c - c0] + O.polar_discriminator(y, 1 / (2 * np.pi))
    local = np.concatenate(outs) if outs else np.zeros(0)
    per = shard.out_count(n, D, shard.MODE_FM)
    counts = [kk * per for (_, kk) in shard.plan_channel_shards(n_ch, world)]
    got = shard.gather_outputs(torch.from_numpy(local.copy()), counts, dist, dst=0)
    if rank == 0:
        want = np.concatenate([O.chain_i16(O.synth_fm_i16(n, fs=1.0e6, ftone=300.0 + 100.0 * c, dev=5000.0), D, h, O.MODE_FM)
                               + O.polar_discriminator(O.fir_decimate(O.frequency_shift(wide, frs[c]), D, h), 1 / (2 * np.pi))
                               for c in range(n_ch)])
        q.put(bool(np.array_equal(got.numpy(), want)))
    dist.barrier()
    dist.destroy_process_group()


def test_channel_shards_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_chan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True
