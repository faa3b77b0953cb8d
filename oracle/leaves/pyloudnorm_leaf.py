"""Restatement of ``pyloudnorm`` (upstream 0.1.1, unpinned in the reference's
``setup.py:40``): RBJ biquad design for the BS.1770 weighting filters and the
float64 numpy reference meter.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Reference call sites: ``audiotools/core/loudness.py:253-260`` (coefficients,
``Meter(rate)._filters``) and ``tests/core/test_loudness.py:13-52``
(``pyln.Meter(...).integrated_loudness`` as the cross-check).

Pinned here by: the ITU-R BS.1770 48 kHz coefficient table (4-digit agreement,
the documented pyloudnorm-vs-ITU gap) and a full-scale 997/1000 Hz sine
reading -3.01 LKFS (tests/test_oracle_leaves.py).
"""
from collections import OrderedDict

import numpy as np
import scipy.signal


class IIRfilter:
    """pyloudnorm.iirfilter.IIRfilter: one RBJ-cookbook biquad."""

    def __init__(self, G, Q, fc, rate, filter_type, passband_gain=1.0):
        self.G = G
        self.Q = Q
        self.fc = fc
        self.rate = rate
        self.filter_type = filter_type
        self.passband_gain = passband_gain
        self.b, self.a = self.generate_coefficients()

    def generate_coefficients(self):
        A = 10 ** (self.G / 40.0)
        w0 = 2.0 * np.pi * (self.fc / self.rate)
        alpha = np.sin(w0) / (2.0 * self.Q)
        cw = np.cos(w0)
        t = self.filter_type
        if t == "high_shelf":
            b0 = A * ((A + 1) + (A - 1) * cw + 2 * np.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * cw)
            b2 = A * ((A + 1) + (A - 1) * cw - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * cw + 2 * np.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * cw)
            a2 = (A + 1) - (A - 1) * cw - 2 * np.sqrt(A) * alpha
        elif t == "low_shelf":
            b0 = A * ((A + 1) - (A - 1) * cw + 2 * np.sqrt(A) * alpha)
            b1 = 2 * A * ((A - 1) - (A + 1) * cw)
            b2 = A * ((A + 1) - (A - 1) * cw - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) + (A - 1) * cw + 2 * np.sqrt(A) * alpha
            a1 = -2 * ((A - 1) + (A + 1) * cw)
            a2 = (A + 1) + (A - 1) * cw - 2 * np.sqrt(A) * alpha
        elif t == "high_pass":
            b0 = (1 + cw) / 2
            b1 = -(1 + cw)
            b2 = (1 + cw) / 2
            a0 = 1 + alpha
            a1 = -2 * cw
            a2 = 1 - alpha
        elif t == "low_pass":
            b0 = (1 - cw) / 2
            b1 = 1 - cw
            b2 = (1 - cw) / 2
            a0 = 1 + alpha
            a1 = -2 * cw
            a2 = 1 - alpha
        elif t == "peaking":
            b0 = 1 + alpha * A
            b1 = -2 * cw
            b2 = 1 - alpha * A
            a0 = 1 + alpha / A
            a1 = -2 * cw
            a2 = 1 - alpha / A
        elif t == "notch":
            b0 = 1
            b1 = -2 * cw
            b2 = 1
            a0 = 1 + alpha
            a1 = -2 * cw
            a2 = 1 - alpha
        else:
            raise ValueError("Invalid filter type", t)
        return np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0

    def apply_filter(self, data):
        return self.passband_gain * scipy.signal.lfilter(self.b, self.a, data)


class Meter:
    """pyloudnorm.meter.Meter (float64 numpy; data is (samples, channels))."""

    def __init__(self, rate, filter_class="K-weighting", block_size=0.400):
        self.rate = rate
        self.filter_class = filter_class
        self.block_size = block_size

    def integrated_loudness(self, data):
        input_data = np.array(data, dtype=np.float64, copy=True)
        if input_data.ndim == 1:
            input_data = input_data[:, None]
        numSamples, numChannels = input_data.shape
        if numSamples < self.block_size * self.rate:
            raise ValueError("Audio must have length greater than the block size.")

        for _, filter_stage in self._filters.items():
            for ch in range(numChannels):
                input_data[:, ch] = filter_stage.apply_filter(input_data[:, ch])

        G = [1.0, 1.0, 1.0, 1.41, 1.41]
        T_g = self.block_size
        Gamma_a = -70.0
        overlap = 0.75
        step = 1.0 - overlap

        T = numSamples / self.rate
        numBlocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
        j_range = np.arange(0, numBlocks)
        z = np.zeros(shape=(numChannels, numBlocks))

        for i in range(numChannels):
            for j in j_range:
                l = int(T_g * (j * step) * self.rate)
                u = int(T_g * (j * step + 1) * self.rate)
                z[i, j] = (1.0 / (T_g * self.rate)) * np.sum(np.square(input_data[l:u, i]))

        with np.errstate(divide="ignore"):
            l = [-0.691 + 10.0 * np.log10(np.sum([G[i] * z[i, j] for i in range(numChannels)]))
                 for j in j_range]

        J_g = [j for j, l_j in enumerate(l) if l_j >= Gamma_a]
        with np.errstate(invalid="ignore", divide="ignore"):
            z_avg_gated = [np.mean([z[i, j] for j in J_g]) if len(J_g) else np.nan
                           for i in range(numChannels)]
            Gamma_r = -0.691 + 10.0 * np.log10(
                np.sum([G[i] * z_avg_gated[i] for i in range(numChannels)])) - 10.0
            J_g = [j for j, l_j in enumerate(l) if (l_j > Gamma_r and l_j > Gamma_a)]
            z_avg_gated = np.nan_to_num(np.array(
                [np.mean([z[i, j] for j in J_g]) if len(J_g) else np.nan
                 for i in range(numChannels)]))
            LUFS = -0.691 + 10.0 * np.log10(
                np.sum([G[i] * z_avg_gated[i] for i in range(numChannels)]))
        return LUFS

    @property
    def filter_class(self):
        return self._filter_class

    @filter_class.setter
    def filter_class(self, value):
        self._filters = OrderedDict()
        self._filter_class = value
        rate = self.rate
        if value == "K-weighting":
            self._filters["high_shelf"] = IIRfilter(4.0, 1 / np.sqrt(2), 1500.0, rate, "high_shelf")
            self._filters["high_pass"] = IIRfilter(0.0, 0.5, 38.0, rate, "high_pass")
        elif value == "Fenton/Lee 1":
            self._filters["high_shelf"] = IIRfilter(5.0, 1 / np.sqrt(2), 1500.0, rate, "high_shelf")
            self._filters["high_pass"] = IIRfilter(0.0, 0.5, 130.0, rate, "high_pass")
            self._filters["peaking"] = IIRfilter(0.0, 1 / np.sqrt(2), 500.0, rate, "peaking")
        elif value == "Fenton/Lee 2":  # upstream: "not yet implemented", K-weighting
            self._filters["high_self"] = IIRfilter(4.0, 1 / np.sqrt(2), 1500.0, rate, "high_shelf")
            self._filters["high_pass"] = IIRfilter(0.0, 0.5, 38.0, rate, "high_pass")
        elif value == "Dash et al.":
            self._filters["high_pass"] = IIRfilter(0.0, 0.375, 149.0, rate, "high_pass")
            self._filters["peaking"] = IIRfilter(-2.93820927, 1.68878655, 1000.0, rate, "peaking")
        elif value == "custom":
            pass
        else:
            raise ValueError("Invalid filter class:", value)
