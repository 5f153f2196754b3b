"""Audio file I/O (third "next" row of SURVEY.md §8f; AudioFile/{Base,I,O}AudioFile.{h,cpp}).  Host-side code: these tests
need no GPU and run in both suites.

Golden files written by the UNMODIFIED reference's OAudioFile (tests/golden/audio/*, make_golden_audio.py) pin
  - the writer: the same samples through hcv_oaudiofile_* must give byte-identical files,
  - the reader: decoding the reference's files must give exactly what the reference's IAudioFile returns.
Where the reference build is available (this container) a wider sweep cross-checks both directions."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_audio import CASES, FRAMES  # noqa: E402  (case table only)

AUDIO = os.path.join(ROOT, "tests", "golden", "audio")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_audio_v1.npz"))


def write_like(A, path, x, ft, fmt, ch, rate, endian, mode, asf):
    f = A.OAudioFile(path, ft, fmt, ch, rate, None if endian < 0 else endian)
    assert f.isOpen() and not f.getIsError()
    xin = x.astype(np.float32) if asf else x
    if mode == 0:
        f.writeInterleaved(xin)
    elif mode == 2:
        a = x.shape[0] // 3
        f.writeInterleaved(xin[:a])
        f.writeInterleaved(xin[a:])
    else:
        for c in range(ch):
            f.seek(0)
            f.writeChannel(np.ascontiguousarray(xin[:, c]), c)
    assert f.getFrames() == x.shape[0]
    f.close()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_writer_is_byte_identical_to_the_reference(gold, tmp_path, case):
    import hisstools_library_amd.audiofile as A
    name, ft, fmt, ch, rate, endian, mode, asf = case
    out = str(tmp_path / name)
    write_like(A, out, gold[name + "_x"], ft, fmt, ch, rate, endian, mode, asf)
    assert open(out, "rb").read() == open(os.path.join(AUDIO, name), "rb").read()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reader_matches_the_reference(gold, case):
    import hisstools_library_amd.audiofile as A
    name, ft, fmt, ch, rate, endian, mode, asf = case
    f = A.IAudioFile(os.path.join(AUDIO, name))
    assert f.isOpen() and not f.getIsError()
    info = gold[name + "_info"]
    got = [f.getFileType(), f.getPCMFormat(), f.getHeaderEndianness(), f.getAudioEndianness(), f.getSamplingRate(), f.getChannels(), f.getFrames(), f.getBitDepth()]
    if fmt == 5 and ft != 3:
        # the reference reads its own AIFC "fl64" as 32-bit floats (IAudioFile.cpp:372-376); here it is float64
        assert f.getPCMFormat() == A.PCMFormat.kAudioFileFloat64 and f.getBitDepth() == 64
        x = f.readInterleaved(FRAMES, np.float64)
        assert np.array_equal(x, gold[name + "_x"])
        return
    assert [float(v) for v in got] == list(info)
    assert np.array_equal(f.readInterleaved(FRAMES, np.float64), gold[name + "_f64"])
    f.seek(0)
    assert np.array_equal(f.readInterleaved(FRAMES, np.float32), gold[name + "_f32"])
    f.seek(5)
    assert f.getPosition() == 5
    for c in range(ch):
        f.seek(5)
        assert np.array_equal(f.readChannel(7, c, np.float64), gold[name + "_f64"][5:12, c])
    # reading past the end delivers zeros (the reference: stale buffer contents)
    f.seek(FRAMES - 2)
    tail = f.readInterleaved(6, np.float64)
    assert np.array_equal(tail[:2], gold[name + "_f64"][-2:]) and not tail[2:].any()
    f.seek(0)
    raw = f.readRaw(FRAMES)
    assert len(raw) == FRAMES * f.getFrameByteCount()


def test_quantisation_rules(tmp_path):
    """Integer formats: round(x * 2^(bits-1)) with two's-complement wrap at +1.0 (the reference discards its clip,
    OAudioFile.cpp:549); WAVE 8-bit is unsigned and clipped (:552-560)."""
    import hisstools_library_amd.audiofile as A
    x = np.array([1.0, -1.0, 0.5, -0.5, 0.25 + 2 ** -17, 0.0], np.float64).reshape(-1, 1)
    p = str(tmp_path / "q.wav")
    f = A.OAudioFile(p, A.FileType.kAudioFileWAVE, A.PCMFormat.kAudioFileInt16, 1, 48000.0)
    f.writeInterleaved(x)
    f.close()
    y = A.IAudioFile(p).readInterleaved(6, np.float64)[:, 0]
    assert list(y) == [-1.0, -1.0, 0.5, -0.5, 0.25, 0.0]
    f = A.OAudioFile(p, A.FileType.kAudioFileWAVE, A.PCMFormat.kAudioFileInt8, 1, 48000.0)
    f.writeInterleaved(np.array([2.0, -2.0, 0.0], np.float64))
    f.close()
    assert list(A.IAudioFile(p).readInterleaved(3, np.float64)[:, 0]) == [127 / 128, -1.0, 0.0]


def test_errors_are_reported_like_the_reference(tmp_path):
    import hisstools_library_amd.audiofile as A
    f = A.IAudioFile(str(tmp_path / "missing.wav"))
    assert not f.isOpen() and f.getErrors() == [A.Error.ERR_FILE_COULDNT_OPEN]
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"RIFFxxxxNOPE" + b"\0" * 64)
    assert A.IAudioFile(str(junk)).getErrors() == [A.Error.ERR_FILE_UNKNOWN_FORMAT]
    short = tmp_path / "short.wav"
    short.write_bytes(b"RIFF")
    assert A.IAudioFile(str(short)).getErrors() == [A.Error.ERR_FILE_BAD_FORMAT]
    # a WAVE file with a compressed format tag
    good = open(os.path.join(AUDIO, "i16_2ch.wav"), "rb").read()
    bad = bytearray(good)
    bad[20] = 2
    adpcm = tmp_path / "adpcm.wav"
    adpcm.write_bytes(bytes(bad))
    assert A.IAudioFile(str(adpcm)).getErrors() == [A.Error.ERR_WAVE_UNSUPPORTED_FORMAT]
    # AIFC with a wrong specification version / unknown compression
    aifc = bytearray(open(os.path.join(AUDIO, "i16_2ch.aifc"), "rb").read())
    v = bytearray(aifc)
    v[20] ^= 0xFF
    p = tmp_path / "v.aifc"
    p.write_bytes(bytes(v))
    assert A.IAudioFile(str(p)).getErrors() == [A.Error.ERR_AIFC_WRONG_VERSION]
    c = bytearray(aifc)
    c[50:54] = b"ulaw"
    p = tmp_path / "c.aifc"
    p.write_bytes(bytes(c))
    assert A.IAudioFile(str(p)).getErrors() == [A.Error.ERR_AIFC_UNSUPPORTED_FORMAT]
    # AIFC whose COMM chunk stops before the compression tag (18..21 bytes): rejected, never classified from stale bytes
    import struct
    t = bytearray(aifc)
    comm = bytes(t).index(b"COMM")
    t[comm + 4:comm + 8] = struct.pack(">I", 20)
    p = tmp_path / "t.aifc"
    p.write_bytes(bytes(t))
    assert A.IAudioFile(str(p)).getErrors() == [A.Error.ERR_FILE_BAD_FORMAT]
    with pytest.raises(TypeError):
        A.IAudioFile(os.path.join(AUDIO, "i16_2ch.wav")).readInterleaved(2, np.int16)
    # unwritable path
    w = A.OAudioFile(str(tmp_path / "no" / "such" / "dir.wav"), A.FileType.kAudioFileWAVE, A.PCMFormat.kAudioFileInt16, 1, 48000.0)
    assert not w.isOpen() and w.getErrors() == [A.Error.ERR_FILE_COULDNT_OPEN]


def test_extra_chunks_and_plain_aiff_are_parsed(tmp_path):
    """A WAVE file with a LIST chunk before "fmt " and an odd-sized chunk before "data"; a hand-made plain AIFF."""
    import struct
    import hisstools_library_amd.audiofile as A
    pcm = struct.pack("<4h", 16384, -16384, 8192, -8192)
    body = b"WAVE" + b"LIST" + struct.pack("<I", 3) + b"abc\0" + b"fmt " + struct.pack("<IHHIIHH", 18, 1, 2, 8000, 32000, 4, 16) + b"\0\0" \
        + b"fact" + struct.pack("<I", 4) + b"\0\0\0\0" + b"data" + struct.pack("<I", len(pcm)) + pcm
    p = tmp_path / "chunks.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    f = A.IAudioFile(str(p))
    assert not f.getIsError() and (f.getChannels(), f.getFrames(), f.getSamplingRate()) == (2, 2, 8000.0)
    assert np.array_equal(f.readInterleaved(2, np.float64), [[0.5, -0.5], [0.25, -0.25]])
    # AIFF: COMM (18 bytes, 80-bit rate 44100 = 0x400EAC44000000000000), SSND with a 2-byte offset
    comm = struct.pack(">hIh", 1, 3, 16) + bytes.fromhex("400EAC44000000000000")
    ssnd = struct.pack(">II", 2, 0) + b"\xAA\xBB" + struct.pack(">3h", 16384, -32768, 1)
    form = b"AIFF" + b"COMM" + struct.pack(">I", 18) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    p = tmp_path / "plain.aif"
    p.write_bytes(b"FORM" + struct.pack(">I", len(form)) + form)
    f = A.IAudioFile(str(p))
    assert not f.getIsError() and f.getFileType() == A.FileType.kAudioFileAIFF and f.getSamplingRate() == 44100.0
    assert list(f.readChannel(3, 0, np.float64)) == [0.5, -1.0, 1 / 32768]


def test_wider_sweep_against_the_reference_build(tmp_path):
    """Every (type, format, channels, endianness, write mode, input precision) combination, both directions."""
    from oracle import oracle as O
    if not O.have_ref_audio():
        pytest.skip("reference build (oracle/_ref) not available here")
    import hisstools_library_amd.audiofile as A
    rng = np.random.default_rng(0)
    checked = 0
    for ft in (3, 2, 1):
        for fmt in range(6):
            for ch in (1, 2, 5):
                for endian in (-1, 1) if ft != 3 else (-1, 0, 1):
                    for mode in (0, 1, 2):
                        asf = bool((fmt + ch + mode) & 1)
                        frames = 1500 if mode == 0 else 41                   # > one 1024-frame work loop of the reference
                        x = rng.uniform(-1.2, 1.2, (frames, ch))
                        ref, mine = str(tmp_path / "ref.bin"), str(tmp_path / "mine.bin")
                        assert O.ref_audio_write(ref, ft, fmt, x, 12345.678 if ft != 3 else 48000.0, endian, mode, asf) == 0
                        write_like(A, mine, x, ft, fmt, ch, 12345.678 if ft != 3 else 48000.0, endian, mode, asf)
                        assert open(ref, "rb").read() == open(mine, "rb").read(), (ft, fmt, ch, endian, mode)
                        if fmt == 5 and ft != 3:
                            continue
                        g = A.IAudioFile(ref)
                        assert np.array_equal(g.readInterleaved(frames, np.float64), O.ref_audio_read(ref, frames, ch)[1])
                        g.seek(7)
                        assert np.array_equal(g.readChannel(20, ch - 1, np.float32), O.ref_audio_read(ref, 20, ch, 7, ch - 1, np.float32)[1])
                        checked += 1
    assert checked > 200
