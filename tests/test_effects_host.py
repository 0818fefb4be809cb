"""Host-side logic of the effect classes (no GPU): parameters handed to the C ABI, the tremolo's table bookkeeping."""
import numpy as np
import pytest

from oracle import effects_oracle as fx


@pytest.fixture()
def pkg():
    import pyaudiodsptools_amd as p
    return p


def test_effect_parameters_follow_the_reference_constructors(pkg):
    assert pkg.CreateSoftClipper().drive == pytest.approx(1.44) and pkg.CreateSoftClipper(2.0).params()[0] == 3.0
    s = pkg.CreateSaturator()
    assert s.saturation_coeff == pytest.approx(0.1) and s.mode == 1 and s.params()[1] == pytest.approx(10 ** 0.1)
    assert pkg.CreateSaturator(-12.0, 3.0, 'soft').params()[2] == 2.0
    with pytest.raises(ValueError):
        pkg.CreateSaturator(mode='medium')
    assert pkg.CreateHardDistortion().linear_limit == 0.8
    assert pkg.CreateVolumeChange(6.0).params() == (pytest.approx(10 ** 0.3), 1.0, 0.0)
    assert pkg.CreateVolumeChange(-3.5, False).params()[1] == 0.0


@pytest.mark.parametrize("fs,depth,lfo,chunk", [(44100, 0.4, 4.5, 4096), (48000, 0.9, 7, 1000), (44100, 0.5, 44100 / 1536, 512),
                                                (44100, 0.3, 20.0, 4096), (96000, 1.0, 0.7, 8192)])
def test_tremolo_table_and_phase_bookkeeping_match_the_oracle(pkg, fs, depth, lfo, chunk):
    pkg.config.initialize(fs, chunk)
    t = pkg.CreateTremolo(depth, lfo)
    o = fx.OracleTremolo(fs, depth, lfo)
    assert t.lfo_length == len(o.table)
    assert np.abs(t.sin_lfo.astype(np.float64) - o.table).max() < 2e-6
    ramp = np.arange(len(o.table), dtype=np.float32)
    probe = fx.OracleTremolo(fs, depth, lfo)
    probe.table = ramp  # applying to ones now returns the table indices
    sizes = [chunk] * 40 + [3, 1, chunk // 2, 7 * chunk]
    for n in sizes:
        want = probe.apply(np.ones(n, np.float32))
        assert t._phase(n) == int(want[0])
    t.reset()
    assert t._phase(chunk) == 0


def test_tremolo_needs_config_and_one_stream(pkg):
    pkg.config.initialize(44100, 512)
    t = pkg.CreateTremolo()
    with pytest.raises(ValueError):
        t.apply(np.zeros((2, 512), np.float32))
    with pytest.raises(ValueError):
        pkg.CreateTremolo(0.4, 1e-4)  # period longer than 2**23 samples


def test_mix_signals_argument_errors(pkg):
    with pytest.raises(IndexError):
        pkg.MixSignals()
    with pytest.raises(ValueError, match="equal in length"):
        pkg.MixSignals(np.zeros(4, np.float32), np.zeros(5, np.float32))


def test_iir_coefficients_match_reference_without_gpu():
    """recursive._rbj (host design, float64) against the coefficients captured from the reference."""
    from conftest import load_golden
    from pyaudiodsptools_amd.recursive import _rbj
    want = load_golden("kat_recursive")["iir_coeffs"]
    got = np.array(_rbj("lowshelf", 250, -6, 44100.0) + _rbj("peak", 1500, 3, 44100.0) + _rbj("highshelf", 6000, -2.5, 44100.0))
    assert np.array_equal(got, want)
