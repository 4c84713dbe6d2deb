"""Randomised parity sweep on the GPU: random filter lengths, decimation factors, stream lengths,
modes, input kinds, kernel choices and shard cuts against the CPU oracle, with the tolerances of
tests/test_gpu_parity.py.  Usage: python tools/fuzz_parity.py [seconds] [seed]
Prints one line per failing case (with the seed that reproduces it) and exits non-zero if any."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sdr_amd as sdr                      # noqa: E402
from oracle import oracle                  # noqa: E402
from parity_checks import check_linear, TWO_PI   # noqa: E402
from parity_checks import check_fm as _check_fm  # noqa: E402


def check_fm(got, want, y, scale, what=""):
    """the bars of include/sdr_hip.h with the angle bound on every output (random taps, noise inputs)"""
    _check_fm(got, want, y, scale, what, good_rel=None)


def random_taps(rng, T, D):
    kind = rng.integers(0, 4)
    if kind == 0:
        return oracle.create_fir_filter(T, min(0.8 / D, 0.9), "hann")
    if kind == 1:
        return oracle.create_fir_filter(T, float(rng.uniform(0.01, 0.9)), "blackman")
    if kind == 2:
        return rng.standard_normal(T) * 10.0 ** rng.uniform(-3, 1)
    h = rng.standard_normal(T) * 10.0 ** rng.uniform(-6, 0, T)   # mixed magnitudes
    return h


def one_case(rng, case_seed):
    T = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 100, 128, 199, 256, 300, 511, 640, 1000, 2047,
                        int(rng.integers(1, 700)), int(rng.integers(1, 700)),
                        int(rng.choice([4096, 8191, 16384]))]))
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 10, 12, 15, 16, 20, 25, 32, 40, int(rng.integers(1, 48))]))
    n = int(rng.choice([0, 1, 3, D, D + 1, T, T + D, 4 * D * 511, 5120 * 3 + 1,
                        int(rng.integers(1, 200_000)), int(rng.integers(1, 200_000))]))
    mode = int(rng.choice([sdr.MODE_IQ, sdr.MODE_FM, sdr.MODE_ENV, sdr.MODE_REAL]))
    kind = int(rng.choice([sdr.IN_I16_IQ, sdr.IN_I16_IQ, sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ,
                           sdr.IN_CF32, sdr.IN_F32_REAL]))
    if T > 2047:
        n = min(n, 40_000)
    if case_seed % 7 == 0 and T <= 128:
        # streams long enough to put several tiles on every workgroup of the persistent kernels (a 1024-output tile
        # per CU: more than 262 144 outputs) -- the round-3 failure of the int8 matrix kernel needed exactly that
        n = 300_000 * D + (case_seed * 2654435761) % (1_500_000 if D > 2 else 900_000)
    mix = float(rng.choice([0.0, 0.0, 0.0371, -0.123]))
    pipe = int(rng.choice([0, 0, 0, 1, 2, 3, 5]))
    matrix_ok = mix == 0.0 and kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ)
    if pipe == 5 and not matrix_ok:
        pipe = 0
    if (mix != 0.0 or kind not in (sdr.IN_I16_IQ, sdr.IN_I16_QI)) and pipe not in (0, 2, 5):
        pipe = 0
    if kind in (sdr.IN_CF32, sdr.IN_F32_REAL) and pipe == 0 and case_seed % 3 == 0:
        pipe = 7                                     # the DMA-filled float tile, pinned (also an automatic choice)
    h = random_taps(rng, T, D)
    scale = float(rng.choice([1 / TWO_PI, 0.75, -1.0]))
    tag = f"seed={case_seed} T={T} D={D} n={n} mode={mode} in={kind} mix={mix} pipe={pipe}"

    # input and the oracle's view of it (complex128 stream)
    if kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI):
        if mode == sdr.MODE_FM and n > 0:
            iq = oracle.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=1900.0)
            iq = (iq + rng.integers(-40, 41, iq.size)).astype(np.int16)
        else:
            iq = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
        x = oracle.unpack_scale_i16(iq)
        raw = iq
        if kind == sdr.IN_I16_QI:
            raw = np.empty_like(iq)
            raw[0::2], raw[1::2] = iq[1::2], iq[0::2]
    elif kind == sdr.IN_I8_IQ:
        if mode == sdr.MODE_FM and n > 0:
            raw = oracle.synth_fm_i16(n, fs=1.0e6, ftone=3000.0, dev=20000.0, amp=100.0)
            raw = (raw + rng.integers(-3, 4, raw.size)).astype(np.int8)
        else:
            raw = rng.integers(-128, 128, 2 * n, dtype=np.int8)
        x = (raw[0::2].astype(np.float64) + 1j * raw[1::2].astype(np.float64)) / 128.0
    elif kind == sdr.IN_CF32:
        raw = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
        x = raw.astype(np.complex128)
    else:
        raw = (rng.standard_normal(n) * 0.3).astype(np.float32)
        x = raw.astype(np.complex128)
    if mix != 0.0:
        x = oracle.frequency_shift(x, mix)
    try:
        plan = sdr.FirPlan(h, D, mode=mode, input=kind, scale=scale, mix_fr=mix, pipeline=pipe)
    except sdr.SdrError as e:
        lds_limit = T == 16384 and D > 28 and (mix != 0.0 or kind not in (sdr.IN_I16_IQ, sdr.IN_I16_QI))
        assert e.status == sdr.capi.EINVAL and (pipe != 0 or lds_limit), tag + f" plan refused: {e}"
        return "refused"
    tag += f" kernel={plan.info()['kernel']}"
    y = oracle.fir_decimate(x, D, h)
    got = plan.execute_host(raw)
    if T > 4096:
        # float32 accumulation over more than 4096 taps sits at the 1e-5 bound itself (error ~ sqrt(T)
        # ulps of the partial sums): these lengths are held to 2e-5 of full scale
        def check_linear(g, w, what=""):
            w = np.asarray(w)
            assert g.shape == w.shape, (what, g.shape, w.shape)
            if w.size:
                fs = max(1.0, float(np.abs(w).max()))
                err = float(np.abs(g.astype(w.dtype) - w).max())
                assert err <= 2e-5 * fs, (what, err, fs)
    else:
        check_linear = globals()["check_linear"]
    if mode == sdr.MODE_IQ:
        check_linear(got, y, tag)
    elif mode == sdr.MODE_ENV:
        check_linear(got, oracle.envelope(y), tag)
    elif mode == sdr.MODE_REAL:
        check_linear(got, np.ascontiguousarray(y.real), tag)
    else:
        want = oracle.polar_discriminator(y, scale)
        if kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ) and mix == 0.0:
            check_fm(got, want, y, scale, tag)
        else:
            assert got.shape == want.shape, tag     # noise input: angles are ill-conditioned; shape only
    # the same stream at a random sample offset inside a device buffer, outputs at a random element
    # offset inside a buffer of sentinels: identical results, nothing written outside [0, cnt)
    cnt = got.size
    if n > 0:
        isz = raw.itemsize * (2 if kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ) else 1)
        osz = 8 if mode == sdr.MODE_IQ else 4
        so, oo = int(rng.integers(0, 8)), int(rng.integers(0, 6))
        hostbuf = np.zeros(so * isz + raw.nbytes + 64, np.uint8)
        hostbuf[so * isz: so * isz + raw.nbytes] = raw.view(np.uint8).reshape(-1)
        d_in = sdr.DevBuf.from_numpy(hostbuf)
        guard = np.full((oo + cnt + 8) * (osz // 4), 0x7FC0DEAD, np.uint32)
        d_out = sdr.DevBuf.from_numpy(guard)
        assert plan.execute_dev(d_in.at(so * isz), n, d_out.at(oo * osz), cnt) == cnt, tag
        sdr.sync()
        back = d_out.to_numpy(np.uint32, guard.size)
        w = osz // 4
        assert np.all(back[: oo * w] == 0x7FC0DEAD) and np.all(back[(oo + cnt) * w:] == 0x7FC0DEAD), tag + f" wrote outside its outputs (so={so} oo={oo})"
        assert np.array_equal(back[oo * w: (oo + cnt) * w], got.view(np.uint32).reshape(-1)), tag + f" differs at pointer offsets so={so} oo={oo}"
    # a random shard cut of the same stream must reproduce the whole
    int_kind = kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ) and mix == 0.0
    per = 2 if kind in (sdr.IN_I16_IQ, sdr.IN_I16_QI, sdr.IN_I8_IQ) else 1      # array elements per sample
    if cnt >= 2 and (int_kind or mode != sdr.MODE_FM):
        ob = int(rng.integers(0, cnt - 1))
        oe = int(rng.integers(ob + 1, cnt + 1))
        lo = max(0, ob * D - (T - 1) - int(rng.integers(0, 3)))
        hi = min(n, (oe + 1) * D + 1 + int(rng.integers(0, 3)))
        eb = 8 if mode == sdr.MODE_IQ else 4
        d_blk = sdr.DevBuf.from_numpy(raw[per * lo: per * hi])
        d_out = sdr.DevBuf((oe - ob) * eb)
        plan.execute_range_dev(d_blk, lo, hi - lo, n, ob, oe, d_out)
        sdr.sync()
        part = d_out.to_numpy(got.dtype, oe - ob)
        whole = got[ob:oe]
        if mode == sdr.MODE_FM:
            fs = abs(scale) * np.pi
            d = np.abs(part.astype(np.float64) - whole.astype(np.float64))
            d = np.minimum(d, np.abs(2 * fs - d))
            mag = np.abs(y)
            pm = np.maximum(np.minimum(mag[:-1], mag[1:])[ob:oe], 1e-300)
            allowed = abs(scale) * 2.0 * 4e-6 * max(1.0, float(mag.max())) / pm + 4e-6 * fs
            assert not (d > allowed).any(), tag + f" shard [{ob},{oe}) fm {d.max()}"
        else:
            fsl = max(1.0, float(np.abs(got).max()))        # full scale of the stream, not of the cut
            assert np.abs(part - whole).max() <= 4e-6 * fsl, tag + f" shard [{ob},{oe}) {np.abs(part - whole).max()} fs {fsl}"
    plan.destroy()
    return "ok"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sdr.init(0)
    t0 = time.time()
    n_ok = n_ref = n_bad = 0
    i = 0
    last = t0
    while time.time() - t0 < budget:
        case_seed = seed0 * 1_000_003 + i
        i += 1
        rng = np.random.default_rng(case_seed)
        try:
            r = one_case(rng, case_seed)
            n_ok += r == "ok"
            n_ref += r == "refused"
        except AssertionError as e:
            n_bad += 1
            print("FAIL", str(e)[:400], flush=True)
        except sdr.SdrError as e:
            if "pipeline 5 needs at least" in str(e) and "resident samples" in str(e):
                n_ref += 1           # documented limit of the explicit choice (automatic plans delegate)
                continue
            n_bad += 1
            print("ERROR", case_seed, str(e)[:300], flush=True)
        if time.time() - last > 30:
            last = time.time()
            print(f"... {i} cases, {n_bad} failures", flush=True)
    print(f"fuzz: {i} cases, {n_ok} ok, {n_ref} refused geometries, {n_bad} failures", flush=True)
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
