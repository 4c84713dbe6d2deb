import os, sys
sys.path.insert(0, '/root/repo')
import sdr_amd
sdr_amd.init(0)
n = 1 << 26
d_in = sdr_amd.DevBuf(n * 4 + 64)
sdr_amd.capi.synth_fm_i16_dev(d_in, n, fs=61.44e6, ftone=250e3, dev=2.0e6, amp=1200.0, noise_lsb=64, seed=3)
for nfft in (64, 128, 256, 512, 1024, 2048, 4096):
    plan = sdr_amd.FftPlan(nfft, nfft // 2, sdr_amd.window("blackman", nfft))
    frames = plan.frame_count(n)
    d_out = sdr_amd.DevBuf(frames * nfft * 4)
    plan.time_dev(d_in, n, d_out, 20)
    ms = min(plan.time_dev(d_in, n, d_out, 10) for _ in range(3))
    print(nfft, "ms", round(ms, 4), "GB/s", round((4.0 * n + 4.0 * nfft * frames) / ms / 1e6))
    d_out.free()
