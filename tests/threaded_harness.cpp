// CPU harness for tests/test_threaded_split.py: runs the product's chunk split / cross-fade
// (demucs_cpp_amd/host/threaded_inference_hip.hpp: threaded_split_apply) with a stand-in chunk
// inference out[s][c][j] = (s+1) * chunk[c][j] + 0.01 * s * (i+1), so that the arithmetic of
// the driver can be compared with oracle/threaded_split.py without a GPU.
//   harness <L> <num_threads> <S> <in.f32 (2,L) planar> <out.f32 (S,2,L) planar>
#include <cstdio>
#include <cstdlib>

#include "threaded_inference_hip.hpp"

int main(int argc, char **argv)
{
    if (argc != 6)
        return 2;
    const long L = atol(argv[1]);
    const int T = atoi(argv[2]), S = atoi(argv[3]);
    std::vector<float> in((size_t)(2 * L));
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(in.data(), sizeof(float), in.size(), f) != in.size())
        return 3;
    fclose(f);
    demucscpp::StereoMatrix audio(L);
    for (long i = 0; i < L; ++i)
        for (int c = 0; c < 2; ++c)
            audio(c, i) = in[(size_t)(c * L + i)];
    demucscpp::StemTensor out = demucscppthreaded::threaded_split_apply(audio, T, S, [&](int i, const demucscpp::StereoMatrix &chunk) {
        demucscpp::StemTensor o(S, chunk.cols());
        for (int s = 0; s < S; ++s)
            for (int c = 0; c < 2; ++c)
                for (int64_t j = 0; j < chunk.cols(); ++j)
                    o(s, c, j) = (float)(s + 1) * chunk(c, j) + 0.01f * (float)s * (float)(i + 1);
        return o;
    });
    std::vector<float> res((size_t)(S * 2 * L));
    for (int s = 0; s < S; ++s)
        for (int c = 0; c < 2; ++c)
            for (long i = 0; i < L; ++i)
                res[(size_t)((s * 2 + c) * L + i)] = out(s, c, i);
    f = fopen(argv[5], "wb");
    if (!f || fwrite(res.data(), sizeof(float), res.size(), f) != res.size())
        return 4;
    fclose(f);
    return 0;
}
