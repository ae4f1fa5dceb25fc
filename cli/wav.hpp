// wav.hpp — minimal RIFF/WAVE I/O for the CLI drivers (the reference uses libnyquist,
// cli-apps/demucs.cpp:21-105; absent here). Reads PCM 8 (unsigned) / 16 / 24 / 32-bit, IEEE float32 / float64 and
// the 8-bit G.711 companded encodings (A-law, mu-law; decode per ITU-T G.711, checked against Python's audioop),
// skips unknown chunks (e.g. the LIST chunk of test/data/gspi_stereo*.wav), 44.1 kHz
// mono (duplicated to both channels, demucs.cpp:56-64) or stereo only; writes stereo
// float32 like the reference (PCM_FLT, demucs.cpp:100-102).
// Other sample rates are rejected with the reference's message (demucs.cpp:30-36) unless DMX_RESAMPLE=1 is set:
// then the track is converted to 44.1 kHz on the GPU (dmx_resample) and the stems are
// converted back and written at the file's own rate.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "demucscpp_hip.hpp"

namespace wavio
{
inline int resample_device()
{
    const char *d = getenv("DMX_DEVICE");
    return d ? atoi(d) : 0;
}

// native_rate (optional): receives the file's sample rate; a rate other than 44.1 kHz is accepted only when the
// caller passes it AND the environment says DMX_RESAMPLE=1. native_frames (optional): receives the frame count of the
// file when the track was converted (-1 otherwise): write_audio_file cuts the stems back to it (the round trip
// ceil(ceil(N L/M) M/L) is up to a few samples longer than N, and a stem must line up with ITS source file - the count
// travels with the caller, not in a process-wide variable, so several files may be in flight).
inline bool load_audio_file(const std::string &filename, demucscpp::StereoMatrix &out, int *native_rate = nullptr,
                            int64_t *native_frames = nullptr)
{
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f)
    {
        std::cerr << "[ERROR] cannot open " << filename << std::endl;
        return false;
    }
    std::vector<uint8_t> b;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        b.resize((size_t)sz);
        if (fread(b.data(), 1, (size_t)sz, f) != (size_t)sz)
        {
            fclose(f);
            return false;
        }
        fclose(f);
    }
    if (b.size() < 12 || memcmp(b.data(), "RIFF", 4) != 0 || memcmp(b.data() + 8, "WAVE", 4) != 0)
    {
        std::cerr << "[ERROR] not a RIFF/WAVE file: " << filename << std::endl;
        return false;
    }
    uint16_t tag = 0, nch = 0, bits = 0;
    uint32_t rate = 0;
    const uint8_t *data = nullptr;
    size_t dataLen = 0;
    size_t pos = 12;
    while (pos + 8 <= b.size())
    {
        uint32_t sz;
        memcpy(&sz, &b[pos + 4], 4);
        if (memcmp(&b[pos], "fmt ", 4) == 0 && pos + 8 + 16 <= b.size())
        {
            memcpy(&tag, &b[pos + 8], 2);
            memcpy(&nch, &b[pos + 10], 2);
            memcpy(&rate, &b[pos + 12], 4);
            memcpy(&bits, &b[pos + 22], 2);
            // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID's first 2 bytes, if the fmt chunk really holds them
            if (tag == 0xFFFE && sz >= 26 && pos + 8 + 26 <= b.size())
                memcpy(&tag, &b[pos + 8 + 24], 2);
        }
        else if (memcmp(&b[pos], "data", 4) == 0)
        {
            data = &b[pos + 8];
            dataLen = std::min((size_t)sz, b.size() - pos - 8);
        }
        pos += 8 + (size_t)sz + (sz & 1);
    }
    if (!data || nch == 0)
    {
        std::cerr << "[ERROR] malformed wav: " << filename << std::endl;
        return false;
    }
    const char *rs = getenv("DMX_RESAMPLE");
    const bool convert = (int)rate != demucscpp::SUPPORTED_SAMPLE_RATE && native_rate && rs && atoi(rs) == 1 && rate > 0;
    if ((int)rate != demucscpp::SUPPORTED_SAMPLE_RATE && !convert)
    {
        std::cerr << "[ERROR] demucs.cpp only supports the following sample rate (Hz): " << demucscpp::SUPPORTED_SAMPLE_RATE
                  << std::endl; // cli-apps/demucs.cpp:30-36
        return false;
    }
    if (native_rate)
        *native_rate = (int)rate;
    if (native_frames)
        *native_frames = -1;
    if (nch != 1 && nch != 2)
    {
        std::cerr << "[ERROR] demucs.cpp only supports mono and stereo audio" << std::endl; // :42-48
        return false;
    }
    // validate the encoding BEFORE bits / 8 is used as a divisor (a 4-bit ADPCM file must not die with SIGFPE)
    if (!((tag == 3 && (bits == 32 || bits == 64)) || (tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) ||
          ((tag == 6 || tag == 7) && bits == 8)))
    {
        std::cerr << "[ERROR] unsupported wav encoding (tag " << tag << ", " << bits << " bits)" << std::endl;
        return false;
    }
    const size_t bps = bits / 8;
    const size_t N = dataLen / (bps * nch);
    out = demucscpp::StereoMatrix((int64_t)N);
    auto sample = [&](size_t idx) -> float {
        const uint8_t *p = data + idx * bps;
        if (tag == 3 && bits == 32)
        {
            float v;
            memcpy(&v, p, 4);
            return v;
        }
        if (tag == 3 && bits == 64)
        {
            double v;
            memcpy(&v, p, 8);
            return (float)v;
        }
        if (tag == 1 && bits == 8)
            return ((float)p[0] - 128.0f) / 128.0f; // 8-bit PCM is unsigned
        if (tag == 7) // mu-law (G.711): sign, 3-bit exponent, 4-bit mantissa of the complemented byte, bias 0x84
        {
            const uint8_t u = (uint8_t)~p[0];
            int t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4);
            t -= 0x84;
            return (float)((u & 0x80) ? -t : t) / 32768.0f;
        }
        if (tag == 6) // A-law (G.711): byte ^ 0x55, sign bit set = positive
        {
            const uint8_t a = p[0] ^ 0x55;
            const int e = (a & 0x70) >> 4;
            int t = (a & 0x0F) << 4;
            t = e == 0 ? t + 8 : (t + 0x108) << (e - 1);
            return (float)((a & 0x80) ? t : -t) / 32768.0f;
        }
        if (tag == 1 && bits == 16)
        {
            int16_t v;
            memcpy(&v, p, 2);
            return (float)v / 32768.0f;
        }
        if (tag == 1 && bits == 24)
        {
            int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
            return (float)v / 8388608.0f;
        }
        if (tag == 1 && bits == 32)
        {
            int32_t v;
            memcpy(&v, p, 4);
            return (float)v / 2147483648.0f;
        }
        return 0.0f;
    };
    std::cout << "Input samples: " << N << std::endl;
    std::cout << "Length in seconds: " << (double)N / rate << std::endl;
    std::cout << "Number of channels: " << nch << std::endl;
    for (size_t i = 0; i < N; ++i)
    {
        float l = sample(i * nch), r = nch == 2 ? sample(i * nch + 1) : l;
        out(0, (int64_t)i) = l;
        out(1, (int64_t)i) = r;
    }
    if (convert)
    {
        const int64_t n441 = dmx_resample_length((int64_t)N, (int)rate, demucscpp::SUPPORTED_SAMPLE_RATE);
        demucscpp::StereoMatrix conv(n441);
        if (dmx_resample(resample_device(), out.data.data(), (int64_t)N, 2, 1, (int)rate, demucscpp::SUPPORTED_SAMPLE_RATE, conv.data.data()) != DMX_OK)
        {
            std::cerr << "[ERROR] sample-rate conversion failed: " << dmx_last_error() << std::endl;
            return false;
        }
        std::cout << "Converted " << rate << " Hz -> " << demucscpp::SUPPORTED_SAMPLE_RATE << " Hz on the GPU: " << n441 << " samples" << std::endl;
        out = std::move(conv);
        if (native_frames)
            *native_frames = (int64_t)N;
    }
    return true;
}

// stereo float32 WAV; `interleaved` = 2*N floats at 44.1 kHz. out_rate != 44100 (a track that was converted on the way
// in): the stem is converted to that rate on the GPU first and cut to native_frames (>= 0: what load_audio_file reported).
inline bool write_audio_file(const float *interleaved, int64_t N, const std::string &filename, int out_rate = 44100,
                             int64_t native_frames = -1)
{
    std::vector<float> conv;
    if (out_rate != demucscpp::SUPPORTED_SAMPLE_RATE)
    {
        const int64_t n2 = dmx_resample_length(N, demucscpp::SUPPORTED_SAMPLE_RATE, out_rate);
        if (n2 < 0)
            return false;
        conv.resize((size_t)(2 * n2));
        if (dmx_resample(resample_device(), interleaved, N, 2, 1, demucscpp::SUPPORTED_SAMPLE_RATE, out_rate, conv.data()) != DMX_OK)
        {
            std::cerr << "[ERROR] sample-rate conversion failed: " << dmx_last_error() << std::endl;
            return false;
        }
        interleaved = conv.data();
        N = native_frames >= 0 ? std::min<int64_t>(n2, native_frames) : n2; // sample-aligned with the source file
    }
    FILE *f = fopen(filename.c_str(), "wb");
    if (!f)
        return false;
    const uint32_t dataBytes = (uint32_t)(N * 2 * 4), rate = (uint32_t)out_rate, byteRate = rate * 8, fmtLen = 16;
    const uint32_t riffLen = 4 + (8 + fmtLen) + (8 + dataBytes);
    const uint16_t tag = 3, nch = 2, align = 8, bits = 32;
    fwrite("RIFF", 1, 4, f);
    fwrite(&riffLen, 4, 1, f);
    fwrite("WAVE", 1, 4, f);
    fwrite("fmt ", 1, 4, f);
    fwrite(&fmtLen, 4, 1, f);
    fwrite(&tag, 2, 1, f);
    fwrite(&nch, 2, 1, f);
    fwrite(&rate, 4, 1, f);
    fwrite(&byteRate, 4, 1, f);
    fwrite(&align, 2, 1, f);
    fwrite(&bits, 2, 1, f);
    fwrite("data", 1, 4, f);
    fwrite(&dataBytes, 4, 1, f);
    fwrite(interleaved, 4, (size_t)(N * 2), f);
    fclose(f);
    return true;
}
} // namespace wavio
