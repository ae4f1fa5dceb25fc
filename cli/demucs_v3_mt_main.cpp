// demucs_v3_mt.cpp.main — the reference's multi-threaded Demucs v3 (hdemucs_mmi) CLI
// (/root/reference/cli-apps/demucs_v3_mt.cpp:108-227): demucs_v3_mt.cpp.main <model file> <wav file> <out dir> <num threads>.
// <num threads> keeps its meaning as the NUMBER OF COARSE CHUNKS of the 0.75 s-overlap split
// (cli-apps/threaded_inference.hpp:29-193), so the output equals the reference's N-thread output for the
// same per-chunk shifts; the chunks themselves run back to back on the GPU (demucs_cpp_amd/host/
// threaded_inference_hip.hpp). Environment: DMX_DEVICE, DMX_SHIFT_OFFSET, DMX_BATCH.
#include <filesystem>
#include <iomanip>

#include "threaded_inference_hip.hpp"
#include "wav.hpp"

using namespace demucscpp;
using namespace demucscpp_v3;

int main(int argc, const char **argv)
{
    if (argc != 5)
    {
        std::cerr << "Usage: " << argv[0] << " <model file> <wav file> <out dir> <num threads>" << std::endl;
        exit(1);
    }
    std::cout << "demucs_v3_mt.cpp (Multi-threaded) driver program (MI355X HIP path)" << std::endl;
    std::string model_file = argv[1], wav_file = argv[2], out_dir = argv[3];
    int num_threads = 1;
    try
    {
        num_threads = std::stoi(argv[4]);
    }
    catch (const std::exception &)
    {
        std::cerr << "Error: <num threads> must be an integer" << std::endl;
        exit(1);
    }
    StereoMatrix audio;
    int native_rate = SUPPORTED_SAMPLE_RATE; // != 44100 only with DMX_RESAMPLE=1 (wav.hpp)
    int64_t native_frames = -1;              // frames of the source file when it was converted
    if (!wavio::load_audio_file(wav_file, audio, &native_rate, &native_frames))
        exit(1);
    demucs_v3_model model;
    auto ret = load_demucs_v3_model(model_file, &model);
    std::cout << "demucs_model_load returned " << (ret ? "true" : "false") << std::endl;
    if (!ret)
    {
        std::cerr << "Error loading model" << std::endl;
        exit(1);
    }
    const int nb_sources = 4;
    std::cout << "Starting Demucs v3 MMI inference" << std::endl;
    StemTensor out = demucscppthreaded_v3::threaded_inference(model, audio, num_threads);
    static const char *names[4] = {"drums", "bass", "other", "vocals"};
    std::filesystem::path p = out_dir;
    std::filesystem::create_directories(p);
    std::vector<float> wave((size_t)(2 * audio.cols()));
    for (int target = 0; target < nb_sources; ++target)
    {
        auto p_target = p / ("target_" + std::to_string(target) + "_" + names[target] + ".wav");
        std::cout << "Writing wav file " << p_target << std::endl;
        for (int64_t i = 0; i < audio.cols(); ++i)
        {
            wave[(size_t)(2 * i)] = out(target, 0, i);
            wave[(size_t)(2 * i + 1)] = out(target, 1, i);
        }
        if (!wavio::write_audio_file(wave.data(), audio.cols(), p_target.string(), native_rate, native_frames))
        {
            std::cerr << "Error writing " << p_target << std::endl;
            exit(1);
        }
    }
    return 0;
}
