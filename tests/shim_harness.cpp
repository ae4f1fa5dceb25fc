// GPU harness for the C++ shim (demucs_cpp_amd/host/demucscpp_hip.hpp), driven by tests/test_gpu_parity.py:
//
//   shim_harness reentrant <model> <n_samples> <threads>
//       the reference's threaded driver calls demucs_inference concurrently from N std::threads on ONE shared
//       const demucs_model (/root/reference/cli-apps/threaded_inference.hpp:105-123). Runs exactly that
//       pattern (each thread its own input) and requires every result to equal, bit for bit, the result of
//       the same call made alone afterwards.
//   shim_harness eigen <model> <n_samples>
//       (built with -DDEMUCSCPP_HIP_WITH_EIGEN against tests/eigen_stub) the Eigen-typed overloads of
//       demucs_inference / demucs_segment_buffers / model_inference (src/model.hpp:569-666) give the same
//       bits as the container-typed ones.
//   shim_harness v3 <dmc3 model> <n_samples>
//       namespace demucscpp_v3: load_demucs_v3_model / demucs_v3_inference / model_v3_inference
//       (src/model.hpp:1396-1414); a v4 loader on the v3 file (and the reverse) fails with "bad magic".
//   shim_harness wasm <model> <n_samples>   (Eigen build only)
//       the call sequence of the reference's wasm glue, src_wasm/demucs.cpp:100-140, kept as it is written there
//       (Eigen::MatrixXf audio(2, N); audio.setZero(); audio(0, i) = left[i]; ...; Eigen::Tensor3dXf out =
//       demucs_inference(model, audio, cb); out(target, 0, i)) against the container-typed call.
// Prints "OK ..." and exits 0 on success.
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "demucscpp_hip.hpp"

using namespace demucscpp;

static StereoMatrix noise(int64_t n, unsigned seed)
{
    StereoMatrix a(n);
    std::mt19937 g(seed);
    std::normal_distribution<float> N(0.f, 0.1f);
    for (auto &v : a.data)
        v = N(g);
    return a;
}

int main(int argc, char **argv)
{
    if (argc < 4)
        return 2;
    const std::string mode = argv[1];
    const int64_t n = atol(argv[3]);
    if (mode == "v3")
    {
        using namespace demucscpp_v3;
        demucs_model wrong;
        if (load_demucs_model(argv[2], &wrong)) // dmc3 is "bad magic" to the v4 loader (src/model_load.cpp:79-102)
        {
            printf("v4 loader accepted a v3 file\n");
            return 1;
        }
        demucs_v3_model m3;
        m3.shift_offset = 1337;
        if (!load_demucs_v3_model(argv[2], &m3))
            return 3;
        StereoMatrix a = noise(n, 11);
        int calls = 0;
        ProgressCallback cb = [&](float, const std::string &) { ++calls; };
        StemTensor r1 = demucs_v3_inference(m3, a, cb);
        StemTensor r2 = demucs_v3_inference(m3, a, ProgressCallback());
        if (r1.S != 4 || r1.data != r2.data || calls == 0)
        {
            printf("MISMATCH v3 track\n");
            return 1;
        }
        demucs_v3_segment_buffers_plain b(2, DMX_SEGMENT_SAMPLES, 4);
        b.mix = noise(DMX_SEGMENT_SAMPLES, 12);
        demucscpp::stft_buffers sb(DMX_SEGMENT_SAMPLES);
        model_v3_inference(m3, b, sb, cb, 0.f, 1.f);
        double e = 0;
        for (float v : b.targets_out.data)
            e += (double)v * v;
        if (!(e > 0))
        {
            printf("EMPTY v3 segment\n");
            return 1;
        }
        printf("OK v3 shim\n");
        return 0;
    }
    demucs_model model;
    model.shift_offset = 1337;
    if (!load_demucs_model(argv[2], &model))
        return 3;
    if (mode == "reentrant")
    {
        const int T = argc > 4 ? atoi(argv[4]) : 4;
        std::vector<StereoMatrix> in;
        for (int t = 0; t < T; ++t)
            in.push_back(noise(n + 1000 * t, 100 + (unsigned)t));
        std::vector<StemTensor> got((size_t)T);
        std::vector<std::thread> th;
        const demucs_model &shared = model;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] { got[(size_t)t] = demucs_inference(shared, in[(size_t)t], ProgressCallback()); });
        for (auto &x : th)
            x.join();
        for (int t = 0; t < T; ++t)
        {
            StemTensor alone = demucs_inference(shared, in[(size_t)t], ProgressCallback());
            if (alone.data.size() != got[(size_t)t].data.size() ||
                std::memcmp(alone.data.data(), got[(size_t)t].data.data(), alone.data.size() * sizeof(float)) != 0)
            {
                printf("MISMATCH thread %d\n", t);
                return 1;
            }
            double s = 0;
            for (float v : alone.data)
                s += (double)v * v;
            if (!(s > 0))
            {
                printf("EMPTY thread %d\n", t);
                return 1;
            }
        }
        printf("OK reentrant %d threads\n", T);
        return 0;
    }
#ifdef DEMUCSCPP_HIP_WITH_EIGEN
    if (mode == "wasm")
    {
        // src_wasm/demucs.cpp:100-140, verbatim in structure: planar float* in, Eigen types in between, planar float* out
        StereoMatrix ref_in = noise(n, 21);
        std::vector<float> left((size_t)n), right((size_t)n);
        for (int64_t i = 0; i < n; ++i)
            left[(size_t)i] = ref_in(0, i), right[(size_t)i] = ref_in(1, i);
        const size_t N = (size_t)n;
        demucscpp::ProgressCallback progressCallback = [](float, std::string) {};
        int nb_out_targets = model.is_4sources ? 4 : 6;
        Eigen::MatrixXf audio(2, N);
        audio.setZero();
        for (size_t i = 0; i < N; ++i)
        {
            audio(0, i) = left[i];
            audio(1, i) = right[i];
        }
        Tensor3dXf target_waveforms = demucs_inference(model, audio, progressCallback);
        StemTensor want = demucs_inference(model, ref_in, ProgressCallback());
        for (int target = 0; target < nb_out_targets; ++target)
            for (size_t i = 0; i < N; ++i)
                if (target_waveforms(target, 0, i) != want(target, 0, (int64_t)i) || target_waveforms(target, 1, i) != want(target, 1, (int64_t)i))
                {
                    printf("MISMATCH wasm sequence\n");
                    return 1;
                }
        printf("OK wasm call sequence\n");
        return 0;
    }
    if (mode == "eigen")
    {
        StereoMatrix a = noise(n, 7);
        Eigen::MatrixXf e(2, n);
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < 2; ++c)
                e(c, i) = a(c, i);
        StemTensor r1 = demucs_inference(model, a, ProgressCallback());
        Tensor3dXf r2 = demucs_inference(model, e, ProgressCallback());
        for (int s = 0; s < 4; ++s)
            for (int c = 0; c < 2; ++c)
                for (int64_t i = 0; i < n; ++i)
                    if (r1(s, c, i) != r2(s, c, i))
                    {
                        printf("MISMATCH track\n");
                        return 1;
                    }
        demucs_segment_buffers_plain b1(2, DMX_SEGMENT_SAMPLES, 4);
        demucs_segment_buffers b2(2, DMX_SEGMENT_SAMPLES, 4); // the reference name: Eigen-typed under the macro (src/model.hpp:569-647)
        StereoMatrix sg = noise(DMX_SEGMENT_SAMPLES, 9);
        b1.mix = sg;
        for (int64_t i = 0; i < DMX_SEGMENT_SAMPLES; ++i)
            for (int c = 0; c < 2; ++c)
                b2.mix(c, i) = sg(c, i);
        stft_buffers sb(DMX_SEGMENT_SAMPLES);
        int calls = 0;
        ProgressCallback cb = [&](float, const std::string &) { ++calls; };
        model_inference(model, b1, sb, cb, 0.f, 1.f);
        model_inference(model, b2, sb, cb, 0.f, 1.f);
        if (calls != 4 || std::memcmp(b1.targets_out.data.data(), b2.targets_out.data(), b1.targets_out.data.size() * sizeof(float)) != 0)
        {
            printf("MISMATCH segment\n");
            return 1;
        }
        printf("OK eigen overloads\n");
        return 0;
    }
#endif
    return 2;
}
