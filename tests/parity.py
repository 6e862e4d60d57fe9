"""Shared parity checks: a candidate (HIP kernel on the GPU, or the same kernel source run
through tests/emu on the CPU) against the f64 oracle.

Stated tolerance (fp32 transform vs f64 reference, BASELINE.json "within a stated float
tolerance"):
  float rows : ||got - want||_2 / ||want||_2 <= 1e-6, and per bin
               |got - want| <= 2e-6 * max(want) (+1e-6 absolute floor)
  complex    : the same on the complex values
  dB (f32)   : compared through power, see check_db
  u8 pixels  : exact on >= 99.9 % of pixels, |diff| <= 1 elsewhere (the reference truncates
               a double; a 1-ulp log difference can move a value across an integer boundary)
"""
import numpy as np

from oracle import oracle as O

ORACLE_MODE = {0: O.MODE_MAG, 1: O.MODE_DB10_U8, 2: O.MODE_DB5_U8_DCFIX, 3: O.MODE_COMPLEX,
               4: O.MODE_MAG_NODC, 5: O.MODE_DB_F64}

REL_L2_TOL = 1e-6
PER_BIN_TOL = 2e-6


def check_float(got, want):
    got = np.asarray(got, dtype=np.complex128 if np.iscomplexobj(got) else np.float64)
    scale = float(np.max(np.abs(want))) if want.size else 1.0
    den = float(np.linalg.norm(want))
    rel = float(np.linalg.norm(got - want)) / den if den > 0 else float(np.linalg.norm(got - want))
    worst = float(np.max(np.abs(got - want))) if want.size else 0.0
    assert rel <= REL_L2_TOL, "relative L2 error %.3e > %.1e" % (rel, REL_L2_TOL)
    assert worst <= PER_BIN_TOL * scale + 1e-6, "per-bin error %.3e > %.3e" % (worst, PER_BIN_TOL * scale + 1e-6)
    return rel, worst


def check_u8(got, want):
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max(initial=0) <= 1, "pixel differs by %d" % diff.max()
    bad = np.count_nonzero(diff)
    assert bad <= max(1, int(1e-3 * want.size)), "%d of %d pixels differ" % (bad, want.size)
    return bad


def check_db(got, want_db, want_mag_nodc):
    """10*log10(p + 1e-20) in f32: compare where the bin is not numerically empty."""
    floor = 1e-5 * float(np.max(want_mag_nodc))
    keep = want_mag_nodc > floor
    err = np.abs(np.asarray(got, dtype=np.float64) - want_db)[keep]
    # d(dB) = 8.69 * d(mag)/mag ; mags above the floor are accurate to ~2e-6*max/floor relative
    assert err.max(initial=0.0) <= 8.69 * PER_BIN_TOL / 1e-5 + 1e-3


def check_mode(got, iq, n, n_frames, hop, flip, mode):
    want = O.rows(iq, n_frames, n, hop=hop, flip=flip, mode=ORACLE_MODE[mode])
    if mode in (1, 2):
        return check_u8(got, want)
    if mode == 5:
        return check_db(got, want, O.rows(iq, n_frames, n, hop=hop, flip=flip, mode=O.MODE_MAG_NODC))
    return check_float(got, want)


def check_mode_shifted(got, iq, n, n_frames, hop, flip, mode, cycles_per_sample, phase0_cycles=0.0):
    """check_mode for the frequency-shifted path (oracle: orc_rows_shifted)."""
    kw = dict(hop=hop, flip=flip)
    want = O.rows_shifted(iq, n_frames, n, cycles_per_sample, phase0_cycles, mode=ORACLE_MODE[mode], **kw)
    if mode in (1, 2):
        return check_u8(got, want)
    if mode == 5:
        return check_db(got, want, O.rows_shifted(iq, n_frames, n, cycles_per_sample, phase0_cycles,
                                                  mode=O.MODE_MAG_NODC, **kw))
    return check_float(got, want)


def check_mode_windowed(got, iq, n, n_frames, hop, flip, mode, window):
    """check_mode for a plan with a taper window (oracle: orc_rows_windowed, x[j] = (-1)^j w[j] u8[j] / 256)."""
    w = np.asarray(window, dtype=np.float32).astype(np.float64)   # the weights the kernel applies are the f32 values
    kw = dict(hop=hop, flip=flip)
    want = O.rows_windowed(iq, n_frames, n, w, mode=ORACLE_MODE[mode], **kw)
    if mode in (1, 2):
        return check_u8(got, want)
    if mode == 5:
        return check_db(got, want, O.rows_windowed(iq, n_frames, n, w, mode=O.MODE_MAG_NODC, **kw))
    return check_float(got, want)
