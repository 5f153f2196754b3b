#!/usr/bin/env python3
"""Generate tests/golden/audio/* and tests/golden/golden_audio_v1.npz with the UNMODIFIED reference's OAudioFile /
IAudioFile (oracle/_ref/libhisstools_ref_audio.so, compiled from /root/reference by oracle/Makefile).  Run in the build
container only:   python tests/golden/make_golden_audio.py

For every case of CASES the reference writes tests/golden/audio/<name> from the float64 samples stored as <name>_x and the
fixture records what the reference's reader returns for that file (<name>_f64 / <name>_f32, interleaved [frames][channels])
plus its header fields (<name>_info: type, format, header endianness, audio endianness, rate, channels, frames, bit depth).
AIFC float64 files are written but not read back: the reference's reader misreads its own "fl64" (IAudioFile.cpp:372-376).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FRAMES = 29
# (name, file type, pcm format, channels, rate, endianness (-1 default), write mode, float input)
CASES = []
for ft, ext in ((3, "wav"), (2, "aifc")):
    for fmt, tag in enumerate(("i8", "i16", "i24", "i32", "f32", "f64")):
        CASES.append((f"{tag}_2ch.{ext}", ft, fmt, 2, 48000.0, -1, 0, False))
CASES += [("i16_be_3ch.wav", 3, 1, 3, 44100.0, 1, 1, False), ("f32_be_1ch.wav", 3, 4, 1, 96000.0, 1, 0, True),
          ("i24_perchannel_3ch.aifc", 2, 2, 3, 88200.0, -1, 1, True), ("i16_aiff_request_1ch.aif", 1, 1, 1, 22050.0, -1, 2, False),
          ("i24_oddbytes_1ch.wav", 3, 2, 1, 44100.0, -1, 2, False)]


def samples(name, channels):
    rng = np.random.default_rng(abs(hash(name)) % (1 << 32) if False else sum(map(ord, name)))
    x = rng.uniform(-1, 1, (FRAMES, channels))
    x[0, 0], x[1, 0], x[2, 0], x[3, 0] = 1.0, -1.0, 0.999999, 0.0       # full scale wraps on integer formats (no clipping, OAudioFile.cpp:549)
    return x


if __name__ == "__main__":
    from oracle import oracle as O
    G = {}
    out_dir = os.path.join(ROOT, "tests", "golden", "audio")
    for name, ft, fmt, ch, rate, endian, mode, asf in CASES:
        x = samples(name, ch)
        path = os.path.join(out_dir, name)
        flags = O.ref_audio_write(path, ft, fmt, x, rate, endian, mode, asf)
        assert flags == 0, (name, flags)
        G[name + "_x"] = x
        rc, info = O.ref_audio_info(path)
        G[name + "_info"] = np.array([info[k] for k in ("file_type", "pcm_format", "header_endianness", "audio_endianness", "sampling_rate", "channels", "frames", "bit_depth")], np.float64)
        if not (fmt == 5 and ft != 3):
            G[name + "_f64"] = O.ref_audio_read(path, FRAMES, ch)[1]
            G[name + "_f32"] = O.ref_audio_read(path, FRAMES, ch, dtype=np.float32)[1]
    path = os.path.join(ROOT, "tests", "golden", "golden_audio_v1.npz")
    np.savez_compressed(path, **G)
    total = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir))
    print(f"wrote {len(CASES)} files ({total} bytes) and {path} ({os.path.getsize(path)} bytes)")
