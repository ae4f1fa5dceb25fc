// demucscpp_hip.hpp — header-only C++17 mirror of the reference's public API for the hot
// path, on top of the C ABI (include/demucs_hip.h). Same names, argument meaning and
// error behaviour as /root/reference/src/model.hpp:
//
//   bool  demucscpp::load_demucs_model(const std::string&, demucs_model*)      :649-650
//   <S,2,N> demucscpp::demucs_inference(const demucs_model&, <2,N>, ProgressCallback) :658-660
//   void  demucscpp::model_inference(const demucs_model&, demucs_segment_buffers&,
//                                    stft_buffers&, ProgressCallback, float, float) :662-666
//
// Eigen is not required: the two tensor types below have exactly the memory image of
// the reference's column-major Eigen::MatrixXf(2,N) and Eigen::Tensor3dXf(S,2,N), so a
// project that has Eigen can wrap them zero-copy with Eigen::Map / Eigen::TensorMap
// (see INTEGRATION.md). Define DEMUCSCPP_HIP_WITH_EIGEN before including this header to
// get overloads that take and return the Eigen types themselves.
#pragma once
#include "demucs_hip.h"

#include <cstdlib>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#ifdef DEMUCSCPP_HIP_WITH_EIGEN
#include <Eigen/Dense>
#include <unsupported/Eigen/CXX11/Tensor>
#endif

namespace demucscpp
{

using ProgressCallback = std::function<void(float, const std::string &)>; // src/model.hpp:17

const int SUPPORTED_SAMPLE_RATE = 44100; // src/dsp.hpp:14
const float SEGMENT_LEN_SECS = 7.8f;     // src/model.hpp:652
const float MAX_SHIFT_SECS = 0.5f;       // src/model.hpp:654
const float OVERLAP = 0.25f;             // src/model.hpp:655

// (2, N) column-major == interleaved stereo; element (c, i) at c + 2*i
struct StereoMatrix
{
    int64_t n = 0;
    std::vector<float> data;
    StereoMatrix() {}
    explicit StereoMatrix(int64_t cols) : n(cols), data((size_t)(2 * cols), 0.0f) {}
    int64_t rows() const { return 2; }
    int64_t cols() const { return n; }
    float &operator()(int c, int64_t i) { return data[(size_t)(c + 2 * i)]; }
    float operator()(int c, int64_t i) const { return data[(size_t)(c + 2 * i)]; }
};

// (S, 2, N) column-major; element (s, c, i) at s + S*(c + 2*i)
struct StemTensor
{
    int S = 0;
    int64_t n = 0;
    std::vector<float> data;
    StemTensor() {}
    StemTensor(int s, int64_t cols) : S(s), n(cols), data((size_t)(s * 2 * cols), 0.0f) {}
    int64_t dimension(int d) const { return d == 0 ? S : (d == 1 ? 2 : n); }
    float &operator()(int s, int c, int64_t i) { return data[(size_t)(s + (int64_t)S * (c + 2 * i))]; }
    float operator()(int s, int c, int64_t i) const { return data[(size_t)(s + (int64_t)S * (c + 2 * i))]; }
};

// weight container: resident in HBM behind an opaque handle (src/model.hpp:285-554)
struct demucs_model
{
    bool is_4sources = true;
    int device = 0;          // HIP device the weights live on (env DMX_DEVICE at load time)
    int shift_offset = -1;   // -1: rand() % 22050 like src/model_apply.cpp:114; else fixed
    int max_batch = 12;      // segments in flight per context (7.8 GB of arena; 3.4 ms per segment vs 4.1 at 4)
    dmx_model *handle = nullptr;
    mutable dmx_ctx *ctx = nullptr; // lazily created, reused across calls (one per model object)
    demucs_model() {}
    demucs_model(const demucs_model &) = delete;
    demucs_model &operator=(const demucs_model &) = delete;
    ~demucs_model()
    {
        if (ctx)
            dmx_ctx_free(ctx);
        if (handle)
            dmx_model_free(handle);
    }
};

// src/model.hpp:649-650. Returns false and reports on stderr exactly when the reference
// loader does (src/model_load.cpp:64-69,97-102,1065-1070,1096-1105), and additionally
// when no HIP device is usable (there is no CPU fallback).
inline bool load_demucs_model(const std::string &model_file, demucs_model *model)
{
    const char *dev = std::getenv("DMX_DEVICE");
    model->device = dev ? std::atoi(dev) : model->device;
    if (const char *so = std::getenv("DMX_SHIFT_OFFSET"))
        model->shift_offset = std::atoi(so);
    if (const char *mb = std::getenv("DMX_BATCH"))
        model->max_batch = std::max(1, std::atoi(mb));
    if (dmx_model_load(model_file.c_str(), model->device, &model->handle) != DMX_OK)
    {
        std::cerr << "load_demucs_model: " << dmx_last_error() << std::endl;
        return false;
    }
    model->is_4sources = dmx_model_n_sources(model->handle) == 4;
    return true;
}

namespace detail
{
inline dmx_ctx *context(const demucs_model &m)
{
    if (!m.ctx && dmx_ctx_create(m.handle, 0, m.max_batch, &m.ctx) != DMX_OK)
    {
        // the reference has no error channel in inference (std::exit(1), src/layers.cpp:98-103)
        std::cerr << "demucs_inference: " << dmx_last_error() << std::endl;
        std::exit(1);
    }
    return m.ctx;
}
struct CbThunk
{
    const ProgressCallback *cb;
};
inline void progress_thunk(float p, const char *msg, void *user)
{
    const ProgressCallback *cb = static_cast<CbThunk *>(user)->cb;
    if (cb && *cb)
        (*cb)(p, std::string(msg));
}
} // namespace detail

// src/model.hpp:658-660, src/model_apply.cpp:60-91
inline StemTensor demucs_inference(const demucs_model &model, const StereoMatrix &full_audio, ProgressCallback cb)
{
    const int S = model.is_4sources ? 4 : 6;
    StemTensor out(S, full_audio.cols());
    detail::CbThunk th{&cb};
    if (dmx_track_infer(detail::context(model), full_audio.data.data(), full_audio.cols(), model.shift_offset, out.data.data(),
                        DMX_LAYOUT_EIGEN, detail::progress_thunk, &th) != DMX_OK)
    {
        std::cerr << "demucs_inference: " << dmx_last_error() << std::endl;
        std::exit(1);
    }
    return out;
}

// segment-level surface; src/model.hpp:569-647 (only the boundary members are kept:
// `mix` in, `targets_out` out - every intermediate lives in the HBM arena)
struct demucs_segment_buffers
{
    int segment_samples;
    StereoMatrix mix;
    StemTensor targets_out;
    demucs_segment_buffers(int /*nb_channels*/, int segment_samples_, int nb_sources)
        : segment_samples(segment_samples_), mix(segment_samples_), targets_out(nb_sources, segment_samples_)
    {
    }
};
struct stft_buffers // kept for signature compatibility (src/dsp.hpp:20-101); the STFT state lives on the GPU
{
    explicit stft_buffers(int /*n_samples*/) {}
};

// src/model.hpp:662-666, src/model_inference.cpp:48-475
inline void model_inference(const demucs_model &model, demucs_segment_buffers &buffers, stft_buffers & /*stft_buf*/,
                            ProgressCallback cb, float current_progress, float segment_progress)
{
    if (buffers.segment_samples != DMX_SEGMENT_SAMPLES)
    {
        std::cerr << "model_inference: segment must be " << DMX_SEGMENT_SAMPLES << " samples" << std::endl;
        std::exit(1);
    }
    if (cb)
        cb(current_progress, "3., apply_model mix shape: (2, " + std::to_string(buffers.segment_samples) + ")");
    if (dmx_segment_infer(detail::context(model), buffers.mix.data.data(), buffers.targets_out.data.data(), DMX_LAYOUT_EIGEN) !=
        DMX_OK)
    {
        std::cerr << "model_inference: " << dmx_last_error() << std::endl;
        std::exit(1);
    }
    if (cb)
        cb(current_progress + segment_progress, "Mask + istft");
}

#ifdef DEMUCSCPP_HIP_WITH_EIGEN
// Exact reference signatures (zero-copy in, one copy out).
inline Eigen::Tensor<float, 3> demucs_inference(const demucs_model &model, const Eigen::MatrixXf &full_audio, ProgressCallback cb)
{
    const int S = model.is_4sources ? 4 : 6;
    Eigen::Tensor<float, 3> out(S, 2, full_audio.cols());
    detail::CbThunk th{&cb};
    if (dmx_track_infer(detail::context(model), full_audio.data(), full_audio.cols(), model.shift_offset, out.data(),
                        DMX_LAYOUT_EIGEN, detail::progress_thunk, &th) != DMX_OK)
    {
        std::cerr << "demucs_inference: " << dmx_last_error() << std::endl;
        std::exit(1);
    }
    return out;
}
#endif

} // namespace demucscpp
