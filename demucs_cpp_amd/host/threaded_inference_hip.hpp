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

// Where the chunks of a track lie: chunk i covers track samples [begin(i), end(i)) and is handed to the model with
// `context` samples of surroundings on either side (the reference's arithmetic: ceilf of a float quotient, :52-53).
struct ChunkLayout
{
    int64_t total = 0, span = 0, context = OVERLAP_SAMPLES;
    int count = 1;
    ChunkLayout(int64_t total_samples, int chunks) : total(total_samples), count(chunks < 1 ? 1 : chunks)
    {
        span = (int64_t)::ceilf((float)total / (float)count);
    }
    int64_t begin(int i) const { return std::min<int64_t>(total, (int64_t)i * span); }
    int64_t end(int i) const { return std::min<int64_t>(total, begin(i) + span); }
    int64_t padded_length(int i) const { return end(i) - begin(i) + 2 * context; }
    // track position of sample j of chunk i's padded signal (may fall outside [0, total))
    int64_t track_index(int i, int64_t j) const { return (int64_t)i * span + j - context; }
};

// The padded signal of chunk i: [ left context | the chunk | right context ]. Left of the first chunk the first track
// sample is repeated, right of the last chunk (and beyond the end of the track) there is silence (:57-96).
inline demucscpp::StereoMatrix cut_chunk(const demucscpp::StereoMatrix &track, const ChunkLayout &lay, int i)
{
    demucscpp::StereoMatrix padded(lay.padded_length(i)); // zero filled
    const int64_t b = lay.begin(i), e = lay.end(i), ctx = lay.context;
    const bool first = i == 0, last = i == lay.count - 1;
    for (int c = 0; c < 2; ++c)
    {
        for (int64_t k = 0; k < ctx; ++k)
        {
            const int64_t src = b - ctx + k;
            if (first)
                padded(c, k) = lay.total > 0 ? track(c, 0) : 0.0f;
            else if (src >= 0)
                padded(c, k) = track(c, src);
        }
        for (int64_t k = 0; k < e - b; ++k)
            padded(c, ctx + k) = track(c, b + k);
        if (!last)
            for (int64_t k = 0; k < ctx && e + k < lay.total; ++k)
                padded(c, e - b + ctx + k) = track(c, e + k);
    }
    return padded;
}

// Cross-fade weights of the recombination: a triangle over one chunk span, normalised to a maximum of 1 (:134-139);
// sample j of a padded chunk output rises along it through the left context, is 1 inside, and falls along its mirror image
// from sample `span` on (:143-171).
struct FadeWindow
{
    std::vector<float> ramp;
    int64_t span, context;
    FadeWindow(int64_t span_, int64_t context_) : ramp((size_t)std::max<int64_t>(span_, 1)), span(span_), context(context_)
    {
        float top = 0.f;
        for (int64_t k = 0; k < span; ++k)
        {
            ramp[(size_t)k] = (float)std::min(k + 1, span - k);
            top = std::max(top, ramp[(size_t)k]);
        }
        for (int64_t k = 0; k < span; ++k)
            ramp[(size_t)k] /= top;
    }
    float at(int64_t k) const { return k >= 0 && k < span ? ramp[(size_t)k] : 0.0f; }
    float weight(int64_t j) const
    {
        if (j < context)
            return at(j);
        if (j >= span)
            return at(span + 2 * context - j - 1);
        return 1.0f;
    }
};

// Running weighted sum of chunk outputs and of their weights. The reference adds a sample's weight once per (target,
// channel) and divides by (sum / (2 S)) at the end (:164-189); both quirks are part of the result's bits and are kept.
class CrossfadeSum
{
  public:
    CrossfadeSum(int sources, int64_t total) : out_(sources, total), wsum_((size_t)total, 0.0f), sources_(sources), total_(total) {}
    void add(const demucscpp::StemTensor &chunk_out, const ChunkLayout &lay, const FadeWindow &fade, int i)
    {
        const int64_t have = chunk_out.dimension(2), n = std::min<int64_t>(lay.span + 2 * lay.context, have);
        for (int64_t j = 0; j < n; ++j)
        {
            const int64_t g = lay.track_index(i, j);
            if (g < 0 || g >= total_)
                continue;
            const float w = fade.weight(j);
            for (int t = 0; t < sources_; ++t)
                for (int ch = 0; ch < 2; ++ch)
                {
                    out_(t, ch, g) += chunk_out(t, ch, j) * w;
                    wsum_[(size_t)g] += w;
                }
        }
    }
    demucscpp::StemTensor finish()
    {
        const float per = 2.0f * (float)sources_;
        for (int64_t g = 0; g < total_; ++g)
            if (wsum_[(size_t)g] > 0)
                for (int t = 0; t < sources_; ++t)
                    for (int ch = 0; ch < 2; ++ch)
                        out_(t, ch, g) /= (wsum_[(size_t)g] / per);
        return std::move(out_);
    }

  private:
    demucscpp::StemTensor out_;
    std::vector<float> wsum_;
    int sources_;
    int64_t total_;
};

// infer(chunk_index, padded chunk (2, n_i)) -> (S, 2, n_i). Chunks are cut, separated and blended one at a time, in
// track order (the blend's float additions happen in the reference's order: chunk by chunk, sample by sample).
template <class Infer>
demucscpp::StemTensor threaded_split_apply(const demucscpp::StereoMatrix &full_audio, int num_threads, int nb_out_sources, Infer infer)
{
    const ChunkLayout lay(full_audio.cols(), num_threads);
    const FadeWindow fade(lay.span, lay.context);
    CrossfadeSum sum(nb_out_sources, lay.total);
    for (int i = 0; i < lay.count; ++i)
        sum.add(infer(i, cut_chunk(full_audio, lay, i)), lay, fade, i);
    return sum.finish();
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
