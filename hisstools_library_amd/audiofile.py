"""Python mirror of HISSTools::IAudioFile / OAudioFile (AudioFile/IAudioFile.h:37-54, OAudioFile.h:14-37) over the C ABI.

Host-side file I/O (WAVE / AIFF / AIFC; int 8/16/24/32, float 32/64) so that real impulse responses can be loaded into
the convolver and results stored.  Same method names and enumeration values as the reference; numpy arrays in and out.
"""
from __future__ import annotations

import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib


class FileType(IntEnum):           # BaseAudioFile.h:20-26
    kAudioFileNone = 0
    kAudioFileAIFF = 1
    kAudioFileAIFC = 2
    kAudioFileWAVE = 3


class PCMFormat(IntEnum):          # BaseAudioFile.h:27-35
    kAudioFileInt8 = 0
    kAudioFileInt16 = 1
    kAudioFileInt24 = 2
    kAudioFileInt32 = 3
    kAudioFileFloat32 = 4
    kAudioFileFloat64 = 5


class Endianness(IntEnum):         # BaseAudioFile.h:36-40
    kAudioFileLittleEndian = 0
    kAudioFileBigEndian = 1


class Error(IntEnum):              # BaseAudioFile.h:47-66
    ERR_NONE = 0
    ERR_MEM_COULD_NOT_ALLOCATE = 1 << 0
    ERR_FILE_ERROR = 1 << 1
    ERR_FILE_COULDNT_OPEN = 1 << 2
    ERR_FILE_BAD_FORMAT = 1 << 3
    ERR_FILE_UNKNOWN_FORMAT = 1 << 4
    ERR_FILE_UNSUPPORTED_PCM_FORMAT = 1 << 5
    ERR_AIFC_WRONG_VERSION = 1 << 6
    ERR_AIFC_UNSUPPORTED_FORMAT = 1 << 7
    ERR_WAVE_UNSUPPORTED_FORMAT = 1 << 8
    ERR_FILE_COULDNT_WRITE = 1 << 9


class BaseAudioFile:
    def __init__(self):
        self.L = _lib.load()
        self.h = None

    def _info(self) -> _lib.AudioFileInfo:
        info = _lib.AudioFileInfo()
        if self.h:
            self.L.hcv_audiofile_get_info(self.h, C.byref(info))
        return info

    def getFileType(self): return FileType(self._info().file_type)
    def getPCMFormat(self): return PCMFormat(self._info().pcm_format)
    def getHeaderEndianness(self): return Endianness(self._info().header_endianness)
    def getAudioEndianness(self): return Endianness(self._info().audio_endianness)
    def getSamplingRate(self): return self._info().sampling_rate
    def getChannels(self): return self._info().channels
    def getFrames(self): return self._info().frames
    def getBitDepth(self): return self._info().bit_depth
    def getByteDepth(self): return self._info().bit_depth // 8
    def getFrameByteCount(self): return self.getChannels() * self.getByteDepth()
    def getErrorFlags(self): return self._info().error_flags
    def getIsError(self): return self.getErrorFlags() != 0
    def getErrors(self): return [e for e in Error if e and self.getErrorFlags() & e]
    def isOpen(self): return bool(self.h and self.L.hcv_audiofile_is_open(self.h))
    def seek(self, position: int = 0): self.h and self.L.hcv_audiofile_seek(self.h, position)
    def getPosition(self): return self.L.hcv_audiofile_position(self.h) if self.h else 0

    def close(self):
        if self.h:
            self.L.hcv_audiofile_close(self.h)
            self.h = None

    def __enter__(self): return self
    def __exit__(self, *exc): self.close()
    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IAudioFile(BaseAudioFile):
    def __init__(self, path: str = ""):
        super().__init__()
        if path:
            self.open(path)

    def open(self, path: str):
        self.close()
        self.h = self.L.hcv_iaudiofile_open(str(path).encode())

    @staticmethod
    def _sample_dtype(dtype):
        dt = np.dtype(dtype)
        if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError(f"IAudioFile: samples are read as float32 or float64, not {dt}")
        return dt

    def readInterleaved(self, numFrames: int, dtype=np.float32):
        """Returns [numFrames][channels]."""
        out = np.zeros((numFrames, self.getChannels()), self._sample_dtype(dtype))
        if out.size:
            if out.dtype == np.float32:
                self.L.hcv_iaudiofile_read_interleaved_f32(self.h, out.ctypes.data_as(_lib.f32p), numFrames)
            else:
                self.L.hcv_iaudiofile_read_interleaved_f64(self.h, out.ctypes.data_as(_lib.f64p), numFrames)
        return out

    def readChannel(self, numFrames: int, channel: int, dtype=np.float32):
        out = np.zeros(numFrames, self._sample_dtype(dtype))
        if out.size:
            if out.dtype == np.float32:
                self.L.hcv_iaudiofile_read_channel_f32(self.h, out.ctypes.data_as(_lib.f32p), numFrames, channel)
            else:
                self.L.hcv_iaudiofile_read_channel_f64(self.h, out.ctypes.data_as(_lib.f64p), numFrames, channel)
        return out

    def readRaw(self, numFrames: int) -> bytes:
        buf = C.create_string_buffer(max(1, numFrames * self.getFrameByteCount()))
        self.L.hcv_iaudiofile_read_raw(self.h, buf, numFrames)
        return buf.raw[: numFrames * self.getFrameByteCount()]


class OAudioFile(BaseAudioFile):
    def __init__(self, path: str = "", type=FileType.kAudioFileWAVE, format=PCMFormat.kAudioFileInt16, channels: int = 1, sr: float = 44100.0, endianness=None):
        super().__init__()
        if path:
            self.open(path, type, format, channels, sr, endianness)

    def open(self, path: str, type, format, channels: int, sr: float, endianness=None):
        self.close()
        self.h = self.L.hcv_oaudiofile_open(str(path).encode(), int(type), int(format), channels, float(sr), -1 if endianness is None else int(endianness))

    def writeInterleaved(self, data):
        """data: [numFrames][channels] (or 1-D for a mono file), float32 or float64."""
        a = np.ascontiguousarray(data)
        if a.dtype != np.float32:
            a = a.astype(np.float64)
        frames = a.size // max(1, self.getChannels())
        if a.dtype == np.float32:
            self.L.hcv_oaudiofile_write_interleaved_f32(self.h, a.ctypes.data_as(_lib.f32p), frames)
        else:
            self.L.hcv_oaudiofile_write_interleaved_f64(self.h, a.ctypes.data_as(_lib.f64p), frames)

    def writeChannel(self, data, channel: int):
        a = np.ascontiguousarray(data)
        if a.dtype != np.float32:
            a = a.astype(np.float64)
        if a.dtype == np.float32:
            self.L.hcv_oaudiofile_write_channel_f32(self.h, a.ctypes.data_as(_lib.f32p), a.size, channel)
        else:
            self.L.hcv_oaudiofile_write_channel_f64(self.h, a.ctypes.data_as(_lib.f64p), a.size, channel)

    def writeRaw(self, data: bytes, numFrames: int):
        self.L.hcv_oaudiofile_write_raw(self.h, data, numFrames)


def load_impulse_responses(path: str, dtype=np.float32):
    """All channels of an audio file as [channels][frames] plus the sampling rate — ready for Convolver.set."""
    with IAudioFile(path) as f:
        if not f.isOpen() or f.getIsError():
            raise IOError(f"{path}: {[e.name for e in f.getErrors()] or 'could not open'}")
        return np.ascontiguousarray(f.readInterleaved(f.getFrames(), dtype).T), f.getSamplingRate()
