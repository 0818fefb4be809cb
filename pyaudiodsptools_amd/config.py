"""Process-global sampling rate / chunk size, read by device constructors at construction time only.

Mirror of the reference's ``pyAudioDspTools/config.py:23-36`` (same names, same semantics):
``config.initialize(44100, 4096)`` before creating devices.  ``use_gpu`` is stored for signature
compatibility; in this package the FFT devices always run on the GPU (there is no CPU path).
"""
import sys

this = sys.modules[__name__]

this.sampling_rate = None
this.chunk_size = None
this.use_gpu = True
this._gpu_available = True


def initialize(sampling_rate, chunk_size, use_gpu=False):
    this.sampling_rate = sampling_rate
    this.chunk_size = chunk_size
    this.use_gpu = use_gpu
