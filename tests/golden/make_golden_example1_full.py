#!/usr/bin/env python3
"""Golden vector of the reference's Example1.py run IN FULL: TestFile16BitMono.wav -> MakeChunks (65
chunks of 4096, 1640 padded zeros) -> CreateLowCutFilter(800).apply per chunk -> CombineChunks (266240 samples; the last
input chunk is never flushed).  Run in the BUILD container only (imports /root/reference); writes
tests/golden/kat_example1_full.npz: the file's PCM (a data file of the reference, needed as input on the GPU box), every
8th output sample, the first and the last output chunk in full, and the sha256 prefix of the merged float32 output
(numpy-version specific: SURVEY records 9c38cf3169419998 for numpy 2.2.6)."""
import contextlib
import hashlib
import io
import os
import sys

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

with contextlib.redirect_stdout(io.StringIO()):
    import pyAudioDspTools as ref
    ref.config.initialize(44100, 4096)
    full = ref.Utility.MonoWavToNumpyFloat(os.path.join(REF, "TestFile16BitMono.wav"))
chunks = ref.MakeChunks(full)
dev = ref.CreateLowCutFilter(800)
outs = [dev.apply(ch) for ch in chunks]
merged = ref.CombineChunks(outs)
pcm = np.round(full * 32768).astype(np.int16)
assert np.array_equal(pcm.astype(np.float32) / 32768, full)
sha = hashlib.sha256(merged.tobytes()).hexdigest()[:16]
np.savez_compressed(os.path.join(HERE, "kat_example1_full.npz"), pcm16=pcm, out_dec8=merged[::8].astype(np.float32),
                    out_first_chunk=outs[0].astype(np.float32), out_last_chunk=outs[-1].astype(np.float32),
                    n_chunks=np.array([len(chunks)]), out_len=np.array([len(merged)]),
                    sha16=np.frombuffer(sha.encode(), dtype=np.uint8), numpy_version=np.frombuffer(np.__version__.encode(), dtype=np.uint8))
print("chunks", len(chunks), "len", len(merged), "sha16", sha, "numpy", np.__version__)
