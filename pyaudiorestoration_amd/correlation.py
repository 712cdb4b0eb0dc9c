"""Mirror of reference util/correlation.py (xcorr :6-13, find_delay :16-39, parabolic :42-46).

`parabolic` is scalar host arithmetic (its device twin lives inside K_track, csrc/track.hip).
`xcorr` is only used on window-sized vectors by the CorrelationTracker / tape-sync tools and
delegates to scipy.signal.correlate exactly like the reference does.
"""
import logging

import numpy as np
import scipy.signal


def xcorr(a, b, mode='full'):
    """Normalized cross correlation returning correlation in range [-1.0, 1.0]"""
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    return scipy.signal.correlate(a, b, mode=mode, method='auto')


def find_delay(a, b, ignore_phase=False, window_name=None):
    """Calculate the delay between 1D signals a and b (windows a and b in place, like the reference)."""
    if window_name:
        a *= scipy.signal.get_window(window_name, len(a))
        b *= scipy.signal.get_window(window_name, len(b))
    res = xcorr(a, b, mode="same")
    if ignore_phase:
        logging.warning("Ignoring phase")
        max_index = np.argmax(np.abs(res))
    else:
        max_index = np.argmax(res)
    i_peak, corr = parabolic(res, max_index)
    logging.debug(f"i_peak {i_peak}")
    return i_peak - len(res) // 2, corr


def parabolic(f, x):
    """Helper function to refine a peak position in an array"""
    xv = 1 / 2. * (f[x - 1] - f[x + 1]) / (f[x - 1] - 2 * f[x] + f[x + 1]) + x
    yv = f[x] - 1 / 4. * (f[x - 1] - f[x + 1]) * (xv - x)
    return xv, yv
