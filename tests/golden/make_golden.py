#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.npz from the REFERENCE's own functions.

Run in the build container only (needs oracle/_ref/libsdrref.so, which oracle/build_ref.sh
compiles from /root/reference):

    python tests/golden/make_golden.py

Every expected output in the file is produced by the reference's code (polyphase_decimate,
polar_discriminator, frequency_shift, decimate_vec, filter_vec, create_freq_rotation and the
main-loop chain of bladerf_apt_demod.cpp:728-754).  Inputs are seeded and stored alongside
(or, for the large FM case, regenerated from the stored parameters).  The file holds data only.
The FFT and the filter designer have no reference source in the tree (ArrayFire / Common):
their vectors come from numpy.fft / the documented definition and are marked "unpinned".
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    O.build()
    assert O.have_ref(), "oracle/_ref/libsdrref.so missing: run `make -C oracle ref`"
    rng = np.random.default_rng(20260515)
    g = {}

    # (1) int16 unpack KAT incl. the edge values (bladerf_apt_demod.cpp:728-732 through the chain
    #     with a 1-tap unity filter, D = 1, mode IQ => y[n] = x[n]/2048)
    kat = np.array([0, 0, 1, -1, -1, 1, 2047, -2048, -2048, 2047, 32767, -32768, -32768, 32767,
                    1234, -4321, -7, 7], np.int16)
    g["unpack_in"] = kat
    g["unpack_out"] = O.ref_chain_i16(kat, 1, np.ones(1), O.MODE_IQ)

    # (2) impulse / step / random through FIR-decimate, N both = 0 and != 0 (mod D)
    cases = []
    for T in (1, 7, 64, 101, 200, 511):
        for D in (1, 5, 10, 15, 20):
            for N in (20 * D * 3, 20 * D * 3 + 7):
                cases.append((T, D, N))
    g["fir_cases"] = np.array(cases, np.int32)
    for ci, (T, D, N) in enumerate(cases):
        h = rng.standard_normal(T)
        kind = ci % 3
        iq = np.zeros(2 * N, np.int16)
        if kind == 0:          # impulse at a position that is not a multiple of D
            iq[2 * (D + 3)] = 2047
            iq[2 * (D + 3) + 1] = -2048
        elif kind == 1:        # step
            iq[0::2] = 1000
            iq[1::2] = -500
        else:
            iq[:] = rng.integers(-2048, 2048, 2 * N, dtype=np.int16)
        g[f"fir{ci}_h"] = h
        g[f"fir{ci}_iq"] = iq
        g[f"fir{ci}_y"] = O.ref_chain_i16(iq, D, h, O.MODE_IQ)
        g[f"fir{ci}_fm"] = O.ref_chain_i16(iq, D, h, O.MODE_FM)
        g[f"fir{ci}_env"] = O.ref_chain_i16(iq, D, h, O.MODE_ENV)

    # (3) BASELINE config 1: FM 1 kHz tone, 1 MS/s, 10 s, 64-tap Hann LPF (/10).  The input is
    #     regenerated from the parameters (oracle.synth_fm_i16); taps are stored (designer is
    #     unpinned, so they are an INPUT here).
    n1 = 10_000_000
    h64 = O.create_fir_filter(64, 0.1, "hann")
    iq1 = O.synth_fm_i16(n1, fs=1.0e6, ftone=1000.0, dev=5000.0, amp=2000.0)
    fm1 = O.ref_chain_i16(iq1, 10, h64, O.MODE_FM)
    g["cfg1_params"] = np.array([n1, 1.0e6, 1000.0, 5000.0, 2000.0, 10])
    g["cfg1_taps"] = h64
    g["cfg1_iq_sha256"] = np.frombuffer(hashlib.sha256(iq1.tobytes()).digest(), np.uint8)
    g["cfg1_iq_head"] = iq1[:512].copy()
    g["cfg1_fm_head"] = fm1[:256].copy()
    g["cfg1_fm_tail"] = fm1[-256:].copy()
    g["cfg1_fm_len"] = np.array([fm1.size])
    g["cfg1_fm_sum"] = np.array([fm1.sum(), np.abs(fm1).sum(), (fm1 * fm1).sum()])
    g["cfg1_fm_stride"] = fm1[::9973].copy()          # 1003 samples spread over the whole output
    env1 = O.ref_chain_i16(iq1[: 2 * 1_000_000], 10, h64, O.MODE_ENV)
    g["cfg1_env_stride"] = env1[::997].copy()

    # (4) envelope of a two-tone (AM-like) signal, T = 101, D = 20 (rx_am_demod constants)
    n4 = 20_000
    t = np.arange(n4)
    sig = 1500.0 * (1.0 + 0.5 * np.cos(2 * np.pi * 0.0007 * t)) * np.exp(2j * np.pi * 0.0031 * t)
    iq4 = np.empty(2 * n4, np.int16)
    iq4[0::2] = np.rint(sig.real)
    iq4[1::2] = np.rint(sig.imag)
    h101 = O.create_fir_filter(101, 0.02, "hann")
    g["am_iq"] = iq4
    g["am_taps"] = h101
    g["am_env"] = O.ref_chain_i16(iq4, 20, h101, O.MODE_ENV)
    g["am_fm"] = O.ref_chain_i16(iq4, 5, h101, O.MODE_FM, scale=0.75)

    # (6) APT two-stage chain (bladerf_apt_demod.cpp:534-574,728-754): 624 kS/s, 200 taps /10,
    #     FM, +2400/62400 shift, 195 taps /15, envelope.  0.1 s of a 2400 Hz AM subcarrier on FM.
    n6 = 62_400
    tt = np.arange(n6) / 624000.0
    sub = (0.6 + 0.4 * np.sin(2 * np.pi * 40.0 * tt)) * np.sin(2 * np.pi * 2400.0 * tt)
    ph = 2 * np.pi * 17000.0 * np.cumsum(sub) / 624000.0
    iq6 = np.empty(2 * n6, np.int16)
    iq6[0::2] = np.rint(1800.0 * np.cos(ph))
    iq6[1::2] = np.rint(1800.0 * np.sin(ph))
    lpf_fm = O.create_fir_filter(200, 20800 / 62400.0, "hann")
    lpf_am = O.create_fir_filter(195, 2400 / 62400.0, "hann")
    x9, x6 = O.ref_apt_chain_i16(iq6, 10, lpf_fm, 2400 / 62400.0, 15, lpf_am)
    g["apt_iq"] = iq6
    g["apt_lpf_fm"] = lpf_fm
    g["apt_lpf_am"] = lpf_am
    g["apt_x6"] = x6
    g["apt_x9"] = x9
    # long-tap flavour (BASELINE config 5: 511 taps, /10 then /15)
    h511a = O.create_fir_filter(511, 20800 / 62400.0, "hann")
    h511b = O.create_fir_filter(511, 2400 / 62400.0, "hann")
    x9b, x6b = O.ref_apt_chain_i16(iq6, 10, h511a, 2400 / 62400.0, 15, h511b)
    g["apt511_h1"] = h511a
    g["apt511_h2"] = h511b
    g["apt511_x6"] = x6b
    g["apt511_x9"] = x9b

    # (7) the small helpers
    r = rng.standard_normal(4000)
    c = rng.standard_normal(3000) + 1j * rng.standard_normal(3000)
    g["fs_real_in"] = r
    g["fs_real_out"] = O.ref_frequency_shift(r, 2400 / 62400.0)
    g["fs_cplx_in"] = c
    g["fs_cplx_out"] = O.ref_frequency_shift(c, -0.0371)
    g["rot_out"] = O.ref_create_freq_rotation(2048, 0.0123)
    for k, rate in enumerate((2.0, 2.5, 3.3, 10.0)):
        g[f"dec{k}_rate"] = np.array([rate])
        g[f"dec{k}_out"] = O.ref_decimate_vec(r, rate)
    for T in (7, 64, 101):
        hh = rng.standard_normal(T)
        g[f"filt{T}_h"] = hh
        g[f"filt{T}_out"] = O.ref_filter_vec(r, hh)
    y = c[:500]
    g["disc_in"] = y
    g["disc_out"] = O.ref_polar_discriminator(y, 1.0 / (2.0 * np.pi))
    # signed-zero corner: first sample exactly zero, all four sign combinations of the second
    z = np.array([0 + 0j, -1 - 2j, 0 + 0j, 1 - 2j, 0 + 0j, -1 + 2j, 0 + 0j, 1 + 2j])
    g["disc_zero_in"] = z
    g["disc_zero_out"] = O.ref_polar_discriminator(z, 1.0 / (2.0 * np.pi))

    # (5) FFT vectors -- UNPINNED (numpy.fft, not the reference): bin-centred and off-bin tones
    nfft = 4096
    n5 = nfft + 3 * 2048
    t5 = np.arange(n5)
    s5 = 1200.0 * np.exp(2j * np.pi * (300.0 / nfft) * t5) + \
        300.0 * np.exp(2j * np.pi * (1000.37 / nfft) * t5)
    iq5 = np.empty(2 * n5, np.int16)
    iq5[0::2] = np.rint(s5.real) + rng.integers(-64, 65, n5)
    iq5[1::2] = np.rint(s5.imag) + rng.integers(-64, 65, n5)
    g["fft_iq"] = iq5
    xs = (iq5[0::2].astype(np.float64) + 1j * iq5[1::2]) / 2048.0
    wb = 0.42 - 0.5 * np.cos(2 * np.pi * np.arange(nfft) / (nfft - 1)) + \
        0.08 * np.cos(4 * np.pi * np.arange(nfft) / (nfft - 1))
    for name, w in (("rect", np.ones(nfft)), ("blackman", wb)):
        rows = []
        for f in range((n5 - nfft) // 2048 + 1):
            X = np.fft.fft(xs[f * 2048: f * 2048 + nfft] * w) / nfft
            rows.append(np.abs(np.fft.fftshift(X)) ** 2)
        g[f"fft_{name}_pow_unpinned"] = np.array(rows)

    out = os.path.join(HERE, "ref_vectors.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB,", len(g), "arrays")


if __name__ == "__main__":
    main()
