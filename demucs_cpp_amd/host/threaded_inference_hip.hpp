// threaded_inference_hip.hpp — the coarse "N big chunks" driver of the reference's *_mt CLIs
// (/root/reference/cli-apps/threaded_inference.hpp:29-193) on top of demucscpp::demucs_inference.
//
// The reference cuts the track into `num_threads` equal chunks with 0.75 s of context on either
// side, runs demucs_inference on each chunk in its own std::thread, and cross-fades the results.
// On the HIP path one GPU context already keeps every CU busy, so the chunks are pushed through
// the GPU one after another (each chunk is itself a batched overlapping-segment run); what is
// kept is the *arithmetic* of the split and of the recombination, so a `_mt` run with N chunks
// produces what the reference's N-thread run produces for the same per-chunk shift offsets:
//
//   segment_length = ceilf(float(L) / float(N))                                   (:52-53)
//   chunk i = [ left context | audio[i*segment_length, end_i) | right context ]   (:57-96)
//       left context  (33075 samples): i == 0 ? copies of sample 0 : audio[start-33075, start)
//       right context (33075 samples): last chunk ? zeros            : audio[end, end+33075)
//   ramp(k) = min(k+1, segment_length-k) / max                                    (:134-139)
//   out[g] += chunk_out[j] * w(j),  g = i*segment_length + j - 33075,             (:143-171)
//       w(j) = ramp(j) for j < 33075; ramp(segment_length + 2*33075 - j - 1) for j >= segment_length; else 1
//   sum_w[g] += w(j) once per (target, channel)  ->  out[g] /= sum_w[g] / (2*S)   (:164-189)
//
// The chunk inference is a template parameter so that the split / recombination can be tested on
// the CPU against oracle/threaded_split.py with a stand-in inference (tests/test_threaded_split.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "demucscpp_hip.hpp"

namespace demucscppthreaded
{
const int SAMPLE_RATE = 44100;
const float OVERLAP = 0.75f;
const int OVERLAP_SAMPLES = (int)::floorf(SAMPLE_RATE * OVERLAP); // 33075

// infer(chunk_index, chunk (2, n_i)) -> (S, 2, n_i)
template <class Infer>
demucscpp::StemTensor threaded_split_apply(const demucscpp::StereoMatrix &full_audio, int num_threads, int nb_out_sources, Infer infer)
{
    using demucscpp::StemTensor;
    using demucscpp::StereoMatrix;
    const int64_t total_length = full_audio.cols();
    const int64_t segment_length = (int64_t)::ceilf((float)total_length / (float)num_threads);
    const int64_t OV = OVERLAP_SAMPLES;

    std::vector<StereoMatrix> segments;
    for (int i = 0; i < num_threads; ++i)
    {
        const int64_t start = std::min<int64_t>(total_length, i * segment_length);
        const int64_t end = std::min<int64_t>(total_length, start + segment_length);
        StereoMatrix segment(end - start + 2 * OV); // zero filled
        for (int64_t k = 0; k < OV; ++k)
            for (int c = 0; c < 2; ++c)
            {
                if (i == 0)
                    segment(c, k) = total_length > 0 ? full_audio(c, 0) : 0.0f;
                else if (start - OV + k >= 0)
                    segment(c, k) = full_audio(c, start - OV + k);
            }
        if (i != num_threads - 1)
            for (int64_t k = 0; k < OV && end + k < total_length; ++k)
                for (int c = 0; c < 2; ++c)
                    segment(c, end - start + OV + k) = full_audio(c, end + k);
        for (int64_t k = 0; k < end - start; ++k)
            for (int c = 0; c < 2; ++c)
                segment(c, OV + k) = full_audio(c, start + k);
        segments.push_back(std::move(segment));
    }

    std::vector<StemTensor> segment_outs;
    for (int i = 0; i < num_threads; ++i)
        segment_outs.push_back(infer(i, segments[(size_t)i]));

    StemTensor final_output(nb_out_sources, total_length);
    std::vector<float> ramp((size_t)std::max<int64_t>(segment_length, 1));
    float rmax = 0.f;
    for (int64_t k = 0; k < segment_length; ++k)
    {
        ramp[(size_t)k] = (float)std::min(k + 1, segment_length - k);
        rmax = std::max(rmax, ramp[(size_t)k]);
    }
    for (int64_t k = 0; k < segment_length; ++k)
        ramp[(size_t)k] /= rmax;
    std::vector<float> sum_weight((size_t)total_length, 0.0f);

    for (size_t i = 0; i < segment_outs.size(); ++i)
    {
        const int64_t segment_start = (int64_t)i * segment_length;
        const int64_t have = segment_outs[i].dimension(2);
        for (int64_t j = 0; j < segment_length + 2 * OV && j < have; ++j)
        {
            const int64_t g = segment_start + j - OV;
            if (g < 0 || g >= total_length)
                continue;
            float weight = 1.0f;
            if (j < OV)
                weight = j < segment_length ? ramp[(size_t)j] : 0.0f;
            else if (j >= segment_length)
            {
                const int64_t r = segment_length + 2 * OV - j - 1;
                weight = (r >= 0 && r < segment_length) ? ramp[(size_t)r] : 0.0f;
            }
            for (int t = 0; t < nb_out_sources; ++t)
                for (int ch = 0; ch < 2; ++ch)
                {
                    final_output(t, ch, g) += segment_outs[i](t, ch, j) * weight;
                    sum_weight[(size_t)g] += weight; // once per (target, channel), like the reference
                }
        }
    }
    const float per = 2.0f * (float)nb_out_sources;
    for (int64_t g = 0; g < total_length; ++g)
        if (sum_weight[(size_t)g] > 0)
            for (int t = 0; t < nb_out_sources; ++t)
                for (int ch = 0; ch < 2; ++ch)
                    final_output(t, ch, g) /= (sum_weight[(size_t)g] / per);
    return final_output;
}

// cli-apps/threaded_inference.hpp:29-33. `num_threads` chunks; progress lines carry the reference's
// "[THREAD i]" prefix. Chunks run back to back on the model's GPU context.
inline demucscpp::StemTensor threaded_inference(const demucscpp::demucs_model &model, const demucscpp::StereoMatrix &full_audio,
                                                int num_threads, const std::string &prefix = "")
{
    if (num_threads < 1)
        num_threads = 1;
    std::cout << std::fixed << std::setprecision(3);
    const int S = model.is_4sources ? 4 : 6;
    return threaded_split_apply(full_audio, num_threads, S, [&](int i, const demucscpp::StereoMatrix &chunk) {
        demucscpp::ProgressCallback cb = [i, prefix](float progress, const std::string &log_message) {
            std::cout << prefix << "[THREAD " << i << "] (" << std::setw(3) << std::setfill(' ') << progress * 100.0f << "%) "
                      << log_message << std::endl;
        };
        return demucscpp::demucs_inference(model, chunk, cb);
    });
}
} // namespace demucscppthreaded

// cli-apps/threaded_inference.hpp:196-369: the same split / cross-fade around demucscpp_v3::demucs_v3_inference
namespace demucscppthreaded_v3
{
inline demucscpp::StemTensor threaded_inference(const demucscpp_v3::demucs_v3_model &model, const demucscpp::StereoMatrix &full_audio,
                                                int num_threads, const std::string &prefix = "")
{
    if (num_threads < 1)
        num_threads = 1;
    std::cout << std::fixed << std::setprecision(3);
    return demucscppthreaded::threaded_split_apply(full_audio, num_threads, 4, [&](int i, const demucscpp::StereoMatrix &chunk) {
        demucscpp::ProgressCallback cb = [i, prefix](float progress, const std::string &log_message) {
            std::cout << prefix << "[THREAD " << i << "] (" << std::setw(3) << std::setfill(' ') << progress * 100.0f << "%) "
                      << log_message << std::endl;
        };
        return demucscpp_v3::demucs_v3_inference(model, chunk, cb);
    });
}
} // namespace demucscppthreaded_v3
