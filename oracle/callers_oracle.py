"""CPU oracle for the callers that embed the FFT filters: CreateDelay and the private reverb.
TEST INFRASTRUCTURE ONLY - see fftfilter_oracle.py.

Both are tapped delay lines: the chunk, scaled by gain k, is added into a float32 accumulation buffer `spacing*(k+1)`
samples ahead; the head of the buffer is the delayed signal.  Restated with the same float32 additions in the same order
as the reference, so the goldens (tests/golden/kat_callers.npz, captured from the reference) match bit for bit.

Reference anchors:
  * delay     pyAudioDspTools/EffectDelay.py:31-74      (gains linspace(0.5, 0.1, loops), taps at T, 2T, ..)
  * reverb    pyAudioDspTools/_EffectReverb.py:6-61     (two lines: HighCut(5000) -> 99 taps, HighCut(150) -> 49 taps, wet)
The reference's delay calls non-existent filter methods when use_lowcut_filter / use_highcut_filter are set
(EffectDelay.py:56,58 -> AttributeError); OracleDelay implements what its reverb's delay line does there instead
(_EffectReverb.py:41-44: filter.apply), which is also what the product does.
"""
import numpy as np

from . import fftfilter_oracle as orc

F = np.float32


class TappedLine:
    def __init__(self, spacing, gains, length, wet):
        self.spacing = int(spacing)
        self.gains = np.asarray(gains, F)
        self.acc = np.zeros(int(length), F)
        self.wet = wet

    def apply(self, x):
        x = np.asarray(x, F)
        n = len(x)
        for k, g in enumerate(self.gains):
            lo = self.spacing * (k + 1)
            self.acc[lo:lo + n] += x * g  # raises like the reference if the buffer is too short for the chunk
        out = self.acc[:n].copy() if self.wet else x + self.acc[:n]
        self.acc = np.concatenate([self.acc[n:], np.zeros(n, F)])
        return out


class OracleDelay:
    def __init__(self, fs, chunk, time_in_ms=500, feedback_loops=2, lowcut_filter_frequency=40,
                 highcut_filter_frequency=12000, use_lowcut_filter=False, use_highcut_filter=False, wet=False):
        t = int(time_in_ms * (fs / 1000))
        self.line = TappedLine(t, np.linspace(0.5, 0.1, num=feedback_loops, dtype=F), t * (feedback_loops + 2), wet)
        self.lowcut = orc.OracleLowCut(lowcut_filter_frequency, fs, chunk) if use_lowcut_filter else None
        self.highcut = orc.OracleHighCut(highcut_filter_frequency, fs, chunk) if use_highcut_filter else None

    def apply(self, x):
        if self.lowcut is not None:
            x = self.lowcut.apply(x)
        if self.highcut is not None:
            x = self.highcut.apply(x)
        return self.line.apply(x)


class OracleReverb:
    def __init__(self, fs, chunk, time_in_ms=1500):
        total = int((time_in_ms / 1000) * fs)
        self.lines = []
        for loops, cutoff in ((100, 5000), (50, 150)):
            gains = np.linspace(0.3, 0.01, num=loops, dtype=F)[:loops - 1]  # the reference's loop stops one short
            self.lines.append((orc.OracleHighCut(cutoff, fs, chunk), TappedLine(total // loops, gains, total, True)))

    def applyreverb(self, x):
        parts = [line.apply(hc.apply(x)) for hc, line in self.lines]
        return parts[0] + parts[1]


def tap_table(spacing, gains):
    """(delay in samples, gain) pairs of a TappedLine - the form the GPU engine takes."""
    return [(int(spacing) * (k + 1), float(g)) for k, g in enumerate(np.asarray(gains, F))]
