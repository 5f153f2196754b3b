// Drop-in for AudioFile/{Base,I,O}AudioFile.h: HISSTools::BaseAudioFile / IAudioFile / OAudioFile with the reference's
// enumerations, method names and signatures, over the C ABI (hisstools_amd.h: hcv_*audiofile_*).  Host-side file I/O.
#pragma once

#include "../hisstools_amd.h"

#include <cstdint>
#include <string>
#include <vector>

namespace HISSTools
{
    class BaseAudioFile
    {
    public:

        typedef uint32_t FrameCount;
        typedef uintptr_t ByteCount;

        enum FileType { kAudioFileNone, kAudioFileAIFF, kAudioFileAIFC, kAudioFileWAVE };
        enum PCMFormat { kAudioFileInt8, kAudioFileInt16, kAudioFileInt24, kAudioFileInt32, kAudioFileFloat32, kAudioFileFloat64 };
        enum Endianness { kAudioFileLittleEndian, kAudioFileBigEndian };
        enum NumberFormat { kAudioFileInt, kAudioFileFloat };
        enum Error
        {
            ERR_NONE = 0,
            ERR_MEM_COULD_NOT_ALLOCATE = 1 << 0,
            ERR_FILE_ERROR = 1 << 1,
            ERR_FILE_COULDNT_OPEN = 1 << 2,
            ERR_FILE_BAD_FORMAT = 1 << 3,
            ERR_FILE_UNKNOWN_FORMAT = 1 << 4,
            ERR_FILE_UNSUPPORTED_PCM_FORMAT = 1 << 5,
            ERR_AIFC_WRONG_VERSION = 1 << 6,
            ERR_AIFC_UNSUPPORTED_FORMAT = 1 << 7,
            ERR_WAVE_UNSUPPORTED_FORMAT = 1 << 8,
            ERR_FILE_COULDNT_WRITE = 1 << 9,
        };

        BaseAudioFile() : mHandle(nullptr) {}
        virtual ~BaseAudioFile() { release(); }
        BaseAudioFile(const BaseAudioFile &) = delete;
        BaseAudioFile &operator=(const BaseAudioFile &) = delete;

        FileType getFileType() const { return static_cast<FileType>(info().file_type); }
        PCMFormat getPCMFormat() const { return static_cast<PCMFormat>(info().pcm_format); }
        Endianness getHeaderEndianness() const { return static_cast<Endianness>(info().header_endianness); }
        Endianness getAudioEndianness() const { return static_cast<Endianness>(info().audio_endianness); }
        double getSamplingRate() const { return info().sampling_rate; }
        uint16_t getChannels() const { return static_cast<uint16_t>(info().channels); }
        FrameCount getFrames() const { return info().frames; }
        uint16_t getBitDepth() const { return static_cast<uint16_t>(info().bit_depth); }
        uint16_t getByteDepth() const { return getBitDepth() / 8; }
        ByteCount getFrameByteCount() const { return static_cast<ByteCount>(getChannels()) * getByteDepth(); }
        NumberFormat getNumberFormat() const { return findNumberFormat(getPCMFormat()); }

        int getErrorFlags() const { return info().error_flags; }
        bool getIsError() const { return getErrorFlags() != ERR_NONE; }
        std::vector<Error> getErrors() const
        {
            std::vector<Error> ret;
            for (int i = 0; i < 16; i++)
                if (getErrorFlags() & (1 << i)) ret.push_back(static_cast<Error>(1 << i));
            return ret;
        }
        static std::string getErrorString(Error error)
        {
            switch (error)
            {
                case ERR_MEM_COULD_NOT_ALLOCATE: return "mem could not allocate";
                case ERR_FILE_ERROR: return "file error";
                case ERR_FILE_COULDNT_OPEN: return "file couldn't open";
                case ERR_FILE_BAD_FORMAT: return "file bad format";
                case ERR_FILE_UNKNOWN_FORMAT: return "file unknown format";
                case ERR_FILE_UNSUPPORTED_PCM_FORMAT: return "file unsupported pcm format";
                case ERR_AIFC_WRONG_VERSION: return "aifc wrong version";
                case ERR_AIFC_UNSUPPORTED_FORMAT: return "aifc unsupported format";
                case ERR_WAVE_UNSUPPORTED_FORMAT: return "wave unsupported format";
                case ERR_FILE_COULDNT_WRITE: return "file couldn't write";
                default: return "no error";
            }
        }

        static uint16_t findBitDepth(PCMFormat f)
        {
            return f == kAudioFileInt8 ? 8 : f == kAudioFileInt24 ? 24 : (f == kAudioFileInt32 || f == kAudioFileFloat32) ? 32 : f == kAudioFileFloat64 ? 64 : 16;
        }
        static NumberFormat findNumberFormat(PCMFormat f) { return (f == kAudioFileFloat32 || f == kAudioFileFloat64) ? kAudioFileFloat : kAudioFileInt; }

        virtual void close() { release(); }
        virtual bool isOpen() { return mHandle && hcv_audiofile_is_open(mHandle); }
        virtual void seek(FrameCount position = 0) { if (mHandle) hcv_audiofile_seek(mHandle, position); }
        virtual FrameCount getPosition() { return mHandle ? hcv_audiofile_position(mHandle) : 0; }

    protected:

        void release()
        {
            if (mHandle) hcv_audiofile_close(mHandle);
            mHandle = nullptr;
        }

        hcv_audiofile_info info() const
        {
            hcv_audiofile_info i = {};
            if (mHandle) hcv_audiofile_get_info(mHandle, &i);
            return i;
        }

        hcv_audiofile *mHandle;
    };

    class IAudioFile : public BaseAudioFile
    {
    public:

        IAudioFile(const std::string &path = std::string()) { open(path); }

        void open(const std::string &path)
        {
            release();
            if (!path.empty()) mHandle = hcv_iaudiofile_open(path.c_str());
        }

        void readRaw(void *output, FrameCount numFrames) { if (mHandle) hcv_iaudiofile_read_raw(mHandle, output, numFrames); }
        void readInterleaved(double *output, FrameCount numFrames) { if (mHandle) hcv_iaudiofile_read_interleaved_f64(mHandle, output, numFrames); }
        void readInterleaved(float *output, FrameCount numFrames) { if (mHandle) hcv_iaudiofile_read_interleaved_f32(mHandle, output, numFrames); }
        void readChannel(double *output, FrameCount numFrames, uint16_t channel) { if (mHandle) hcv_iaudiofile_read_channel_f64(mHandle, output, numFrames, channel); }
        void readChannel(float *output, FrameCount numFrames, uint16_t channel) { if (mHandle) hcv_iaudiofile_read_channel_f32(mHandle, output, numFrames, channel); }
    };

    class OAudioFile : public BaseAudioFile
    {
    public:

        OAudioFile() {}
        OAudioFile(const std::string &path, FileType type, PCMFormat format, uint16_t channels, double sr) { open(path, type, format, channels, sr); }
        OAudioFile(const std::string &path, FileType type, PCMFormat format, uint16_t channels, double sr, Endianness e) { open(path, type, format, channels, sr, e); }

        void open(const std::string &path, FileType type, PCMFormat format, uint16_t channels, double sr)
        {
            release();
            mHandle = hcv_oaudiofile_open(path.c_str(), type, format, channels, sr, -1);
        }

        void open(const std::string &path, FileType type, PCMFormat format, uint16_t channels, double sr, Endianness e)
        {
            release();
            mHandle = hcv_oaudiofile_open(path.c_str(), type, format, channels, sr, e);
        }

        void writeInterleaved(const double *input, FrameCount numFrames) { if (mHandle) hcv_oaudiofile_write_interleaved_f64(mHandle, input, numFrames); }
        void writeInterleaved(const float *input, FrameCount numFrames) { if (mHandle) hcv_oaudiofile_write_interleaved_f32(mHandle, input, numFrames); }
        void writeChannel(const double *input, FrameCount numFrames, uint16_t channel) { if (mHandle) hcv_oaudiofile_write_channel_f64(mHandle, input, numFrames, channel); }
        void writeChannel(const float *input, FrameCount numFrames, uint16_t channel) { if (mHandle) hcv_oaudiofile_write_channel_f32(mHandle, input, numFrames, channel); }
        void writeRaw(const char *input, FrameCount numFrames) { if (mHandle) hcv_oaudiofile_write_raw(mHandle, input, numFrames); }
    };
}
