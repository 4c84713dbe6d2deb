"""Multi-GPU sharding plans for the receive chain: host logic only.

The path shards without any exchange step.  Two plans:
  * independent channels (BASELINE config 3): block-partition the channels over the ranks;
  * one long stream (configs 1/2/4): partition the OUTPUTS into contiguous ranges; rank r keeps
    its input range plus the FIR history halo resident (T-1 samples on the left; one more output,
    i.e. D samples, on the right for the discriminator) and runs sdr_fir_execute_range_dev.
The only collective is the final gather of the decimated outputs (RCCL over xGMI on GPUs, gloo
in the CPU tests).  No reference counterpart: the reference is single-process
(bladerf/rx_fm_demod/src/bladerf_fm_demod.cpp:281-333).
"""
from dataclasses import dataclass

MODE_IQ, MODE_FM, MODE_ENV, MODE_REAL = 0, 1, 2, 3


COUNT_FLOOR, COUNT_AF_SEQ = 0, 1


def out_count(n_samples, decim, mode, count_rule=COUNT_FLOOR):
    """floor(N/D) outputs (bladerf_apt_demod.cpp:282) -- or, with COUNT_AF_SEQ, every index 0, D, ... <= N-1
    as x3(af::seq(0, N, D)) picks them (bladerf_fm_demod.cpp:212,294; sdr_hip.h sdr_count_rule) -- one
    fewer for the discriminator (:409)."""
    if count_rule == COUNT_AF_SEQ:
        ny = (n_samples - 1) // decim + 1 if n_samples > 0 else 0
    else:
        ny = n_samples // decim
    if mode == MODE_FM:
        return max(ny - 1, 0)
    return ny


@dataclass(frozen=True)
class TimeShard:
    rank: int
    out_begin: int      # first final-output index of this rank
    out_end: int        # one past the last
    lo: int             # first resident sample (stream index)
    hi: int             # one past the last resident sample

    @property
    def n_out(self):
        return self.out_end - self.out_begin

    @property
    def n_resident(self):
        return self.hi - self.lo


def plan_time_shards(n_total, decim, n_taps, mode, world, count_rule=COUNT_FLOOR):
    """Contiguous output ranges with the input window (own samples + halo) each rank must hold."""
    cnt = out_count(n_total, decim, mode, count_rule)
    shards = []
    for r in range(world):
        ob, oe = cnt * r // world, cnt * (r + 1) // world
        if oe > ob:
            lo = max(0, ob * decim - (n_taps - 1))
            last_y = oe - 1 + (1 if mode == MODE_FM else 0)
            hi = min(n_total, last_y * decim + 1)
        else:
            lo = hi = 0
        shards.append(TimeShard(r, ob, oe, lo, hi))
    return shards


def plan_channel_shards(n_channels, world):
    """[(first_channel, n_channels_of_rank)] -- block partition, remainder to the low ranks."""
    base, rem = divmod(n_channels, world)
    out, c = [], 0
    for r in range(world):
        k = base + (1 if r < rem else 0)
        out.append((c, k))
        c += k
    return out


def plan_frame_shards(n_total, nfft, hop, world):
    """FFT frames are independent: contiguous frame ranges; a rank needs samples
    [f0*hop, (f1-1)*hop + nfft).  Returns [(f0, f1, lo, hi)]."""
    frames = 0 if n_total < nfft else (n_total - nfft) // hop + 1
    out = []
    for r in range(world):
        f0, f1 = frames * r // world, frames * (r + 1) // world
        lo, hi = (f0 * hop, (f1 - 1) * hop + nfft) if f1 > f0 else (0, 0)
        out.append((f0, f1, lo, hi))
    return out


def gather_outputs(local, counts, dist, dst=0, device=None):
    """Final gather of the per-rank outputs on rank `dst` (torch.distributed: nccl == RCCL on ROCm,
    gloo on CPU).  `local` is a 1-D tensor holding this rank's outputs; counts[r] their number on
    rank r.  Returns the concatenated tensor on dst, None elsewhere."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    m = max(counts)
    send = torch.zeros(m, dtype=local.dtype, device=local.device)
    send[: counts[rank]] = local[: counts[rank]]
    recv = [torch.empty(m, dtype=local.dtype, device=local.device) for _ in range(world)] \
        if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    return torch.cat([recv[r][: counts[r]] for r in range(world)])
