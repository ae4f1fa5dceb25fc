// demucs_v3.cpp.main — Demucs v3 (hdemucs_mmi) CLI with the reference's argv contract and output naming
// (/root/reference/cli-apps/demucs_v3.cpp:108-232): demucs_v3.cpp.main <model file> <wav file> <out dir>
// -> <out dir>/target_{i}_{drums|bass|other|vocals}.wav (stereo float32). The model file carries the "dmc3" magic
// (ggml-model-hdemucs_mmi-v3-f16.bin).
// Environment: DMX_DEVICE (GPU index), DMX_SHIFT_OFFSET (fixed shift instead of rand()),
// DMX_BATCH (segments in flight).
#include <filesystem>
#include <iomanip>

#include "wav.hpp"

using namespace demucscpp;
using namespace demucscpp_v3;

int main(int argc, const char **argv)
{
    if (argc != 4)
    {
        std::cerr << "Usage: " << argv[0] << " <model file> <wav file> <out dir>" << std::endl;
        exit(1);
    }
    std::cout << "demucs_v3.cpp Main driver program (MI355X HIP path)" << std::endl;
    std::string model_file = argv[1], wav_file = argv[2], out_dir = argv[3];
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
    std::cout << std::fixed << std::setprecision(3);
    ProgressCallback cb = [](float progress, const std::string &msg) {
        std::cout << "(" << std::setw(3) << std::setfill(' ') << progress * 100.0f << "%) " << msg << std::endl;
    };
    StemTensor out = demucs_v3_inference(model, audio, cb);
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
