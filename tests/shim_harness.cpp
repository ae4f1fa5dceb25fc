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
    demucs_model model;
    model.shift_offset = 1337;
    if (!load_demucs_model(argv[2], &model))
        return 3;
    const int64_t n = atol(argv[3]);
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
