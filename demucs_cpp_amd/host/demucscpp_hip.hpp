// demucscpp_hip.hpp — header-only C++17 mirror of the reference's public API for the hot
// path, on top of the C ABI (include/demucs_hip.h). Same names, argument meaning and
// error behaviour as /root/reference/src/model.hpp:
//
//   bool  demucscpp::load_demucs_model(const std::string&, demucs_model*)      :649-650
//   <S,2,N> demucscpp::demucs_inference(const demucs_model&, <2,N>, ProgressCallback) :658-660
//   void  demucscpp::model_inference(const demucs_model&, demucs_segment_buffers&,
//                                    stft_buffers&, ProgressCallback, float, float) :662-666
// and, for Demucs v3 (hdemucs_mmi), namespace demucscpp_v3 (src/model.hpp:668-1415):
//   bool  load_demucs_v3_model(const std::string&, demucs_v3_model*)                     :1396-1397
//   <4,2,N> demucs_v3_inference(const demucs_v3_model&, <2,N>, ProgressCallback)          :1405-1408
//   void  model_v3_inference(const demucs_v3_model&, demucs_v3_segment_buffers&, stft_buffers&, ...) :1410-1414
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
#include <algorithm>
#include <memory>
#include <mutex>
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

// weight container: resident in HBM (on every device of the engine) behind an opaque handle
// (src/model.hpp:285-554). One engine = weights + activation arena(s) + streams; see include/demucs_hip.h.
//
// Re-entrancy: the reference's contract is concurrent demucs_inference calls on ONE shared const model
// (cli-apps/threaded_inference.hpp:105-123 runs N std::threads on it). Calls on one demucs_model are
// serialised here by `lock` (one GPU context already keeps every CU busy, so nothing is lost); results are
// bit-identical to sequential calls. tests/threaded_harness.cpp runs exactly that pattern.
struct engine_model // what a loaded model is on this side of the boundary, whatever its architecture
{
    std::vector<int> devices;  // HIP devices (env DMX_DEVICES="0,1,..."; "all"; default: device DMX_DEVICE or 0)
    int shift_offset = -1;     // -1: rand() % 22050 like src/model_apply.cpp:114; else fixed
    int max_batch = 12;        // segments in flight per device (7.8 GB of arena; 3.33 ms per segment, 3.9 at 4, 3.22 at 24: DMX_BATCH)
    dmx_engine *engine = nullptr;
    mutable std::mutex lock;
    engine_model() {}
    engine_model(const engine_model &) = delete;
    engine_model &operator=(const engine_model &) = delete;
    ~engine_model()
    {
        if (engine)
            dmx_engine_free(engine);
    }
};
struct demucs_model : engine_model
{
    bool is_4sources = true;
};

namespace detail
{
// DMX_DEVICES: comma-separated HIP device ids, or "all"; DMX_DEVICE: one id (kept from round 1)
inline std::vector<int> devices_from_env()
{
    std::vector<int> d;
    if (const char *e = std::getenv("DMX_DEVICES"))
    {
        std::string s(e);
        if (s == "all")
        {
            for (int i = 0; i < dmx_device_count(); ++i)
                d.push_back(i);
            return d;
        }
        std::stringstream ss(s);
        std::string tok;
        while (std::getline(ss, tok, ','))
            if (!tok.empty())
                d.push_back(std::atoi(tok.c_str()));
    }
    if (d.empty())
    {
        const char *one = std::getenv("DMX_DEVICE");
        d.push_back(one ? std::atoi(one) : 0);
    }
    return d;
}
inline void read_env(int &shift_offset, int &max_batch)
{
    if (const char *so = std::getenv("DMX_SHIFT_OFFSET"))
        shift_offset = std::atoi(so);
    if (const char *mb = std::getenv("DMX_BATCH"))
        max_batch = std::max(1, std::atoi(mb));
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
[[noreturn]] inline void die(const char *where)
{
    // the reference has no error channel in inference (std::exit(1), src/layers.cpp:98-103)
    std::cerr << where << ": " << dmx_last_error() << std::endl;
    std::exit(1);
}
} // namespace detail

// src/model.hpp:649-650. Returns false and reports on stderr exactly when the reference
// loader does (src/model_load.cpp:64-69,97-102,1065-1070,1096-1105), and additionally
// when no HIP device is usable (there is no CPU fallback).
namespace detail
{
// arch: 4 = dmc4 / dmc6 file (HTDemucs v4), 3 = dmc3 (Demucs v3). A file of the other family is "bad magic" to the
// reference's loaders (src/model_load.cpp:79-102 / :1335-1340).
inline bool load_engine(const char *who, const std::string &model_file, engine_model *model, int arch)
{
    if (model->devices.empty())
        model->devices = devices_from_env();
    read_env(model->shift_offset, model->max_batch);
    const char *files[1] = {model_file.c_str()};
    if (dmx_engine_create(files, 1, model->devices.data(), (int)model->devices.size(), model->max_batch, DMX_TRANSPORT_AUTO,
                          &model->engine) != DMX_OK)
    {
        std::cerr << who << ": " << dmx_last_error() << std::endl;
        return false;
    }
    if (dmx_engine_arch(model->engine) != arch)
    {
        std::cerr << who << ": invalid model data (bad magic)" << std::endl;
        dmx_engine_free(model->engine);
        model->engine = nullptr;
        return false;
    }
    return true;
}
} // namespace detail
inline bool load_demucs_model(const std::string &model_file, demucs_model *model)
{
    if (!detail::load_engine("load_demucs_model", model_file, model, 4))
        return false;
    model->is_4sources = dmx_engine_n_sources(model->engine) == 4;
    return true;
}

// src/model.hpp:658-660, src/model_apply.cpp:60-91. With several devices the overlapping-segment loop is
// sharded over them (csrc/engine.cpp); the result is bit-identical to one device.
inline StemTensor demucs_inference(const demucs_model &model, const StereoMatrix &full_audio, ProgressCallback cb)
{
    const int S = model.is_4sources ? 4 : 6;
    StemTensor out(S, full_audio.cols());
    detail::CbThunk th{&cb};
    std::lock_guard<std::mutex> guard(model.lock);
    if (dmx_engine_track_infer(model.engine, full_audio.data.data(), full_audio.cols(), &model.shift_offset, out.data.data(),
                               DMX_LAYOUT_EIGEN, detail::progress_thunk, &th) != DMX_OK)
        detail::die("demucs_inference");
    return out;
}

// The fine-tuned bag (cli-apps/demucs_ft.cpp:136-241): four 4-source models, stem i from model i. The
// reference runs four demucs_inference calls back to back; calling demucs_inference on four demucs_model
// objects still works here, but one bag engine deals all (model, segment) items over the devices at once
// (168 items on 8 GPUs = 21 each, instead of four rounds of 42 = 6 + 6 + 5 + ...).
struct demucs_ft_bag
{
    std::vector<int> devices;
    int shift_offsets[4] = {-1, -1, -1, -1}; // -1: successive rand() % 22050 draws, like four demucs_inference calls
    int max_batch = 12;
    dmx_engine *engine = nullptr;
    mutable std::mutex lock;
    demucs_ft_bag() {}
    demucs_ft_bag(const demucs_ft_bag &) = delete;
    demucs_ft_bag &operator=(const demucs_ft_bag &) = delete;
    ~demucs_ft_bag()
    {
        if (engine)
            dmx_engine_free(engine);
    }
};
// model_files: drums, bass, other, vocals (the order of cli-apps/demucs_ft.cpp:141-168)
inline bool load_demucs_ft_bag(const std::vector<std::string> &model_files, demucs_ft_bag *bag)
{
    if (model_files.size() != 4)
    {
        std::cerr << "load_demucs_ft_bag: four model files are required" << std::endl;
        return false;
    }
    if (bag->devices.empty())
        bag->devices = detail::devices_from_env();
    int so = -1;
    detail::read_env(so, bag->max_batch);
    if (so >= 0)
        for (int &v : bag->shift_offsets)
            v = so;
    const char *files[4] = {model_files[0].c_str(), model_files[1].c_str(), model_files[2].c_str(), model_files[3].c_str()};
    if (dmx_engine_create(files, 4, bag->devices.data(), (int)bag->devices.size(), bag->max_batch, DMX_TRANSPORT_AUTO, &bag->engine) !=
        DMX_OK)
    {
        std::cerr << "load_demucs_ft_bag: " << dmx_last_error() << std::endl;
        return false;
    }
    return true;
}
inline StemTensor demucs_ft_inference(const demucs_ft_bag &bag, const StereoMatrix &full_audio, ProgressCallback cb)
{
    StemTensor out(4, full_audio.cols());
    detail::CbThunk th{&cb};
    std::lock_guard<std::mutex> guard(bag.lock);
    if (dmx_engine_track_infer(bag.engine, full_audio.data.data(), full_audio.cols(), bag.shift_offsets, out.data.data(), DMX_LAYOUT_EIGEN,
                               detail::progress_thunk, &th) != DMX_OK)
        detail::die("demucs_ft_inference");
    return out;
}

// segment-level surface; src/model.hpp:569-647 (only the boundary members are kept:
// `mix` in, `targets_out` out - every intermediate lives in the HBM arena)
// Under DEMUCSCPP_HIP_WITH_EIGEN the reference's name `demucs_segment_buffers` IS the Eigen-typed struct further
// down (a caller that keeps `demucscpp::demucs_segment_buffers buffers(2, n, S); buffers.mix(i, j) = ...` with Eigen
// semantics, src/model.hpp:569-647, must get Eigen members); this container-typed one is then reachable as
// demucs_segment_buffers_plain only.
struct demucs_segment_buffers_plain
{
    int segment_samples;
    StereoMatrix mix;
    StemTensor targets_out;
    demucs_segment_buffers_plain(int /*nb_channels*/, int segment_samples_, int nb_sources)
        : segment_samples(segment_samples_), mix(segment_samples_), targets_out(nb_sources, segment_samples_)
    {
    }
};
#ifndef DEMUCSCPP_HIP_WITH_EIGEN
typedef demucs_segment_buffers_plain demucs_segment_buffers;
#endif
struct stft_buffers // kept for signature compatibility (src/dsp.hpp:20-101); the STFT state lives on the GPU
{
    explicit stft_buffers(int /*n_samples*/) {}
};

namespace detail
{
inline void segment_call(const engine_model &model, int segment_samples, const float *mix, float *targets_out, const ProgressCallback &cb,
                         float current_progress, float segment_progress)
{
    if (segment_samples != DMX_SEGMENT_SAMPLES)
    {
        std::cerr << "model_inference: segment must be " << DMX_SEGMENT_SAMPLES << " samples" << std::endl;
        std::exit(1);
    }
    if (cb)
        cb(current_progress, "3., apply_model mix shape: (2, " + std::to_string(segment_samples) + ")");
    {
        std::lock_guard<std::mutex> guard(model.lock);
        dmx_ctx *ctx = dmx_engine_root_ctx(model.engine, 0);
        if (!ctx || dmx_segment_infer(ctx, mix, targets_out, DMX_LAYOUT_EIGEN) != DMX_OK)
            die("model_inference");
    }
    if (cb)
        cb(current_progress + segment_progress, "Mask + istft");
}
} // namespace detail

// src/model.hpp:662-666, src/model_inference.cpp:48-475
inline void model_inference(const demucs_model &model, demucs_segment_buffers_plain &buffers, stft_buffers & /*stft_buf*/,
                            ProgressCallback cb, float current_progress, float segment_progress)
{
    detail::segment_call(model, buffers.segment_samples, buffers.mix.data.data(), buffers.targets_out.data.data(), cb, current_progress,
                         segment_progress);
}

#ifdef DEMUCSCPP_HIP_WITH_EIGEN
// Exact reference signatures on the Eigen types themselves (zero-copy in, written in place out): a
// caller that keeps /root/reference/src/model.hpp:569-666 as it is compiles against these.
typedef Eigen::Tensor<float, 3> Tensor3dXf; // src/tensor.hpp

inline Tensor3dXf demucs_inference(const demucs_model &model, const Eigen::MatrixXf &full_audio, ProgressCallback cb)
{
    const int S = model.is_4sources ? 4 : 6;
    Tensor3dXf out(S, 2, full_audio.cols());
    detail::CbThunk th{&cb};
    std::lock_guard<std::mutex> guard(model.lock);
    if (dmx_engine_track_infer(model.engine, full_audio.data(), full_audio.cols(), &model.shift_offset, out.data(), DMX_LAYOUT_EIGEN,
                               detail::progress_thunk, &th) != DMX_OK)
        detail::die("demucs_inference");
    return out;
}

// src/model.hpp:569-647: the boundary members with the reference's names and types. The reference's
// intermediates (x, xt, saved_*, ... :583-647) live in the HBM arena and have no host image.
struct demucs_segment_buffers // the reference's name and member types
{
    int segment_samples;
    Eigen::MatrixXf mix;     // (nb_channels, segment_samples)
    Tensor3dXf targets_out;  // (nb_sources, nb_channels, segment_samples)
    demucs_segment_buffers(int nb_channels, int segment_samples_, int nb_sources)
        : segment_samples(segment_samples_), mix(nb_channels, segment_samples_), targets_out(nb_sources, nb_channels, segment_samples_)
    {
        mix.setZero();
        targets_out.setZero();
    }
};
typedef demucs_segment_buffers demucs_segment_buffers_eigen; // round-2 name
inline void model_inference(const demucs_model &model, demucs_segment_buffers &buffers, stft_buffers & /*stft_buf*/,
                            ProgressCallback cb, float current_progress, float segment_progress)
{
    detail::segment_call(model, buffers.segment_samples, buffers.mix.data(), buffers.targets_out.data(), cb, current_progress,
                         segment_progress);
}
#endif

} // namespace demucscpp

// ---------------------------------------------------------------------------------------------
// Demucs v3 (hdemucs_mmi): /root/reference/src/model.hpp:668-1415, cli-apps/demucs_v3.cpp. Same engine, same
// threading contract; the weight file carries the "dmc3" magic (README.md:82 ggml-model-hdemucs_mmi-v3-f16.bin).
namespace demucscpp_v3
{
using demucscpp::ProgressCallback;
using demucscpp::StemTensor;
using demucscpp::StereoMatrix;

struct demucs_v3_model : demucscpp::engine_model // src/model.hpp:694-1236: the weights live in HBM
{
};

// src/model.hpp:1396-1397, src/model_load.cpp:1302-2166
inline bool load_demucs_v3_model(const std::string &model_file, demucs_v3_model *model)
{
    return demucscpp::detail::load_engine("load_demucs_v3_model", model_file, model, 3);
}

// src/model.hpp:1405-1408, src/model_apply.cpp:307-339
inline StemTensor demucs_v3_inference(const demucs_v3_model &model, const StereoMatrix &full_audio, ProgressCallback cb)
{
    StemTensor out(4, full_audio.cols());
    demucscpp::detail::CbThunk th{&cb};
    std::lock_guard<std::mutex> guard(model.lock);
    if (dmx_engine_track_infer(model.engine, full_audio.data.data(), full_audio.cols(), &model.shift_offset, out.data.data(),
                               DMX_LAYOUT_EIGEN, demucscpp::detail::progress_thunk, &th) != DMX_OK)
        demucscpp::detail::die("demucs_v3_inference");
    return out;
}

// src/model.hpp:1238-1394: the boundary members (`mix` in, `targets_out` out); LSTM state, decay tables and every
// intermediate live in the HBM arena
typedef demucscpp::demucs_segment_buffers_plain demucs_v3_segment_buffers_plain;
#ifndef DEMUCSCPP_HIP_WITH_EIGEN
typedef demucscpp::demucs_segment_buffers_plain demucs_v3_segment_buffers;
#endif

// src/model.hpp:1410-1414, src/model_inference.cpp:477-856
inline void model_v3_inference(const demucs_v3_model &model, demucs_v3_segment_buffers_plain &buffers, demucscpp::stft_buffers & /*stft_buf*/,
                               ProgressCallback cb, float current_progress, float segment_progress)
{
    demucscpp::detail::segment_call(model, buffers.segment_samples, buffers.mix.data.data(), buffers.targets_out.data.data(), cb,
                                    current_progress, segment_progress);
}

#ifdef DEMUCSCPP_HIP_WITH_EIGEN
inline demucscpp::Tensor3dXf demucs_v3_inference(const demucs_v3_model &model, const Eigen::MatrixXf &full_audio, ProgressCallback cb)
{
    demucscpp::Tensor3dXf out(4, 2, full_audio.cols());
    demucscpp::detail::CbThunk th{&cb};
    std::lock_guard<std::mutex> guard(model.lock);
    if (dmx_engine_track_infer(model.engine, full_audio.data(), full_audio.cols(), &model.shift_offset, out.data(), DMX_LAYOUT_EIGEN,
                               demucscpp::detail::progress_thunk, &th) != DMX_OK)
        demucscpp::detail::die("demucs_v3_inference");
    return out;
}
typedef demucscpp::demucs_segment_buffers demucs_v3_segment_buffers; // Eigen-typed mix / targets_out
inline void model_v3_inference(const demucs_v3_model &model, demucs_v3_segment_buffers &buffers, demucscpp::stft_buffers & /*stft_buf*/,
                               ProgressCallback cb, float current_progress, float segment_progress)
{
    demucscpp::detail::segment_call(model, buffers.segment_samples, buffers.mix.data(), buffers.targets_out.data(), cb, current_progress,
                                    segment_progress);
}
#endif
} // namespace demucscpp_v3
