"""CPU oracle for the recursive devices (last item): IIR 3-band EQ and compressor.
TEST INFRASTRUCTURE ONLY - see fftfilter_oracle.py.  Pinned bit-exactly to tests/golden/kat_recursive.npz (captured from
the reference by tests/golden/make_golden.py).

Reference anchors:
  * EffectEQ3Band.py:31-93   RBJ-cookbook coefficients (Fs hard-wired to 44100), :95-181 the three per-sample loops
  * EffectCompressor.py:26-40 envelopes, :43-125 the attack / hold / release loop nest
  * EffectGate.py:25-40 envelopes (sampling rate hard-wired to 44100), :42-126 the same loop nest on a depth-scaled copy

Quirks restated on purpose:
  * every band is a biquad fed with the input delayed by ONE sample: the reference prepends three old input samples but
    two old output samples, so index i of the recursion meets x[i-1], x[i-2], x[i-3] (EffectEQ3Band.py:109-114);
  * the recursion is evaluated in float64, left to right, and stored into a float32 array (the chunk's dtype);
  * compressor: no hold time, the sample after a completed release passes untouched, a re-trigger during release jumps
    straight to full compression, and `full_envelope` / `counter_freeze` are per-call locals.
"""
import numpy as np

F = np.float32


def rbj_coefficients(low_hz, low_db, mid_hz, mid_db, high_hz, high_db, fs=44100.0):
    """(b0, b1, b2, a0, a1, a2) per band: low shelf (Q 1), peaking (Q 2.5), high shelf (Q 1)."""
    out = {}
    a = np.sqrt(10 ** (low_db / 20))
    w = 2 * np.pi * low_hz / fs
    al = np.sin(w) / 2 * np.sqrt((a + 1 / a) * (1 / 1.0 - 1) + 2)
    cw, rt = np.cos(w), 2 * np.sqrt(a) * al
    out["low"] = (a * ((a + 1) - (a - 1) * cw + rt), 2 * a * ((a - 1) - (a + 1) * cw), a * ((a + 1) - (a - 1) * cw - rt),
                  (a + 1) + (a - 1) * cw + rt, -2 * ((a - 1) + (a + 1) * cw), (a + 1) + (a - 1) * cw - rt)
    a = np.sqrt(10 ** (mid_db / 20))
    w = 2 * np.pi * mid_hz / fs
    al = np.sin(w) / (2 * 2.5)
    out["mid"] = (1 + al * a, -2 * np.cos(w), 1 - al * a, 1 + al / a, -2 * np.cos(w), 1 - al / a)
    a = np.sqrt(10 ** (high_db / 20))
    w = 2 * np.pi * high_hz / fs
    al = np.sin(w) / 2 * np.sqrt((a + 1 / a) * (1 / 1.0 - 1) + 2)
    cw, rt = np.cos(w), 2 * np.sqrt(a) * al
    out["high"] = (a * ((a + 1) + (a - 1) * cw + rt), -2 * a * ((a - 1) + (a + 1) * cw), a * ((a + 1) + (a - 1) * cw - rt),
                   (a + 1) - (a - 1) * cw + rt, 2 * ((a - 1) - (a + 1) * cw), (a + 1) - (a - 1) * cw - rt)
    return out


def normalised(coeffs):
    """The five ratios the recursion uses: b0/a0, b1/a0, b2/a0, a1/a0, a2/a0 (float64)."""
    b0, b1, b2, a0, a1, a2 = coeffs
    return np.array([b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0], dtype=np.float64)


class OracleBiquadBand:
    """y[i] = f32( c0 x[i-1] + c1 x[i-2] + c2 x[i-3] - c3 y[i-1] - c4 y[i-2] ), float64 arithmetic left to right."""

    def __init__(self, coeffs):
        self.c = normalised(coeffs)
        self.x_hist = np.zeros(3, F)  # x[-3], x[-2], x[-1]
        self.y_hist = np.zeros(2, F)  # y[-2], y[-1]

    def apply(self, chunk):
        x = np.concatenate([self.x_hist, np.asarray(chunk, F)])
        y = np.concatenate([self.y_hist, np.zeros(len(chunk), F)])
        c0, c1, c2, c3, c4 = self.c
        for i in range(len(chunk)):
            # x[i + 2] is the input one sample before output i; y[i + 1], y[i] the two previous outputs
            acc = c0 * x[i + 2] + c1 * x[i + 1] + c2 * x[i] - c3 * y[i + 1] - c4 * y[i]
            y[i + 2] = acc
        self.x_hist = x[-3:].copy()
        self.y_hist = y[-2:].copy()
        return y[2:]


class OracleEQ3Band:
    def __init__(self, low_hz, low_db, mid_hz, mid_db, high_hz, high_db):
        co = rbj_coefficients(low_hz, low_db, mid_hz, mid_db, high_hz, high_db)
        self.coefficients = co
        self.low, self.mid, self.high = (OracleBiquadBand(co[k]) for k in ("low", "mid", "high"))

    def applylowband(self, x):
        return self.low.apply(x)

    def applymidband(self, x):
        return self.mid.apply(x)

    def applyhighband(self, x):
        return self.high.apply(x)


RESTING, ATTACK, RELEASE = 0, 1, 2


class OracleCompressor:
    """The reference's loop nest as one state machine that consumes at most one sample per step."""

    def __init__(self, fs, threshold_in_db=-15, ratio=0.60, attack_in_ms=3.1, release_in_ms=30.1):
        self.threshold = F(10 ** (threshold_in_db / 20))
        self.attack = np.linspace(1.0, ratio, num=int((fs / 1000) * attack_in_ms), dtype=F)
        self.release = np.linspace(ratio, 1.0, num=int((fs / 1000) * release_in_ms), dtype=F)
        self.x = self.y = 0
        self.state = RESTING

    def _prepare(self, chunk):
        """(working copy, threshold mask) - EffectCompressor.py:57: the compressor works on the samples as they are."""
        v = np.array(chunk, F)
        return v, np.abs(v) > self.threshold

    def apply(self, chunk):
        v, above = self._prepare(chunk)
        n, x_max, y_max = len(v), len(self.attack), len(self.release)
        full, freeze = True, False  # per-call locals in the reference
        i, where = 0, "top"
        while i < n:
            if where == "top":
                if above[i] or self.x != 0 or self.y != 0:
                    if full and self.state == RESTING:
                        self.x, self.state = 0, ATTACK
                    if not full and self.state == RELEASE:
                        self.x = x_max - int(self.y * (x_max / y_max))
                        freeze, self.state = False, ATTACK
                    where = "attack"
                else:
                    i += 1
            elif where == "attack":
                if self.x < x_max and self.state == ATTACK:
                    v[i] *= self.attack[self.x]
                    i += 1
                    self.x += 1
                else:
                    where = "hold"
            elif where == "hold":
                if above[i] and self.state == ATTACK:
                    v[i] *= self.attack[x_max - 1]
                    i += 1
                else:
                    self.state = RELEASE
                    where = "release"
            else:  # release
                interrupted = False
                if self.y < y_max and self.state == RELEASE:
                    self.x = 0
                    if not above[i]:
                        v[i] *= self.release[self.y]
                        i += 1
                        self.y += 1
                        continue
                    full, self.y, freeze, interrupted = False, 0, True, True
                if self.y == y_max:
                    full, self.state, self.x, self.y = True, RESTING, 0, 0
                if not freeze:
                    i += 1  # the sample after a completed release passes untouched
                where = "top"
        return v


class OracleGate(OracleCompressor):
    """EffectGate.py:6-126: the compressor's loop nest (line for line the same, :61-124) run on `input * depth`
    (:59, a fresh float32 array for float32 input), with the threshold mask taken from the UNSCALED input (:58) and
    envelopes linspace(1, 1/depth) / linspace(1/depth, 1) at a hard-wired 44100 Hz (:29-33).  apply() returns the
    scaled-and-shaped copy (:126); the caller's array is left alone."""

    def __init__(self, threshold_in_db=-5, depth=0.1, attack=3.1, release=200.1):
        self.depth = depth
        self.threshold = F(10 ** (threshold_in_db / 20))
        self.attack = np.linspace(1.0, 1.0 / depth, num=int((44100 / 1000) * attack), dtype=F)
        self.release = np.linspace(1.0 / depth, 1.0, num=int((44100 / 1000) * release), dtype=F)
        self.x = self.y = 0
        self.state = RESTING

    def _prepare(self, chunk):
        raw = np.asarray(chunk, F)
        return raw * self.depth, np.abs(raw) > self.threshold  # float32 array * Python float -> float32 (NEP 50)
