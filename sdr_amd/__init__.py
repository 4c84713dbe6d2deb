"""sdr_amd -- MI355X-native receive-side DSP hot path of davemers0160/SDR.

The product is libsdrhip.so (sdr_amd/csrc/*.hip behind include/sdr_hip.h); this package is the
thin Python host side (ctypes) used by tests and bench.py.
"""
from . import capi  # noqa: F401
from .capi import (FirPlan, FftPlan, FirStream, AptImage, DevBuf, SdrError, MODE_IQ, MODE_FM, MODE_ENV, MODE_REAL,  # noqa: F401
                   IN_I16_IQ, IN_CF32, IN_F32_REAL, IN_I16_QI, IN_I8_IQ, init, sync, window, create_fir_filter)
