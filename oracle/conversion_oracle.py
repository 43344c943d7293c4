"""CPU ORACLE (test infrastructure, NOT product code) for the conversion driver's integer
framing: a plain restatement of ``compound`` and of the window arithmetic in
/root/reference/test.py:46-84 and 92-132.  Pure integer indexing, so parity with the product's
closed-form version (speech-cloner_amd/conversion.py) must be bit-exact.  Pinned by the algebraic
property stated in SURVEY.md section 8a (row a24): the stitched output has exactly N*T frames and
every output frame t comes from a window that covers t.  Only tests/ may import this module."""
import numpy as np


def compound(y0, y1):
    """test.py:46-84, statement by statement."""
    n_quarter = y0.shape[1] // 4
    i_0 = 1
    i_1 = 0
    y_v = [y0[0, :-n_quarter, :]]
    while True:
        do_break = True
        if i_1 < y1.shape[0]:
            y_v.append(y1[i_1, n_quarter:-n_quarter, :])
            i_1 += 1
            do_break = False
        if i_0 < y0.shape[0] - 1:
            y_v.append(y0[i_0, n_quarter:-n_quarter, :])
            i_0 += 1
            do_break = False
        if do_break:
            break
    y_v.append(y0[-1, n_quarter:, :])
    return np.concatenate(y_v, axis=0)


def window_plan(n_frames, sample_rate, hop, n_times, t_s, t_e):
    """test.py:92-119: returns (padded frame count, n_s, n_e)."""
    total = n_frames
    if n_frames % n_times != 0:
        total = n_frames + (n_times - (n_frames % n_times))
    n_hop_s = t_s * sample_rate // hop
    n_hop_e = t_e * sample_rate // hop
    n_hop_e = min(n_hop_e, total)
    n_delta = n_times * ((n_hop_e - n_hop_s) // n_times)
    return total, n_hop_s, n_hop_s + n_delta
