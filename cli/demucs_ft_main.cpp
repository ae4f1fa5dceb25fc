// demucs_ft.cpp.main — bag-of-4 fine-tuned CLI (/root/reference/cli-apps/demucs_ft.cpp:107-309):
// demucs_ft.cpp.main <model dir> <wav file> <out dir>. The directory is scanned for file names
// containing htdemucs_ft_{drums,bass,other,vocals} (:136-184); the four models run as ONE bag
// (each draws its own shift, :221-231; (model, segment) items dealt over DMX_DEVICES) and stem i is taken from model i (:238-241).
#include <array>
#include <filesystem>
#include <iomanip>

#include "wav.hpp"

using namespace demucscpp;

int main(int argc, const char **argv)
{
    if (argc != 4)
    {
        std::cerr << "Usage: " << argv[0] << " <model dir> <wav file> <out dir>" << std::endl;
        exit(1);
    }
    std::cout << "demucs_ft.cpp Main driver program (MI355X HIP path)" << std::endl;
    std::string model_dir = argv[1], wav_file = argv[2], out_dir = argv[3];
    StereoMatrix audio;
    int native_rate = SUPPORTED_SAMPLE_RATE; // != 44100 only with DMX_RESAMPLE=1 (wav.hpp)
    int64_t native_frames = -1;              // frames of the source file when it was converted
    if (!wavio::load_audio_file(wav_file, audio, &native_rate, &native_frames))
        exit(1);
    // One bag engine instead of four demucs_model objects: the (model, segment) items of all four models are
    // dealt over the devices of DMX_DEVICES together (csrc/engine.cpp); each model still draws its own shift
    // (demucs_ft.cpp:221-231 -> model_apply.cpp:114) and stem i is taken from model i (:238-241).
    static const char *keys[4] = {"htdemucs_ft_drums", "htdemucs_ft_bass", "htdemucs_ft_other", "htdemucs_ft_vocals"};
    static const char *names[4] = {"drums", "bass", "other", "vocals"};
    std::vector<std::string> files(4);
    bool have[4] = {false, false, false, false};
    for (const auto &entry : std::filesystem::directory_iterator(model_dir))
        for (int i = 0; i < 4; ++i)
            if (!have[i] && entry.path().string().find(keys[i]) != std::string::npos)
            {
                files[(size_t)i] = entry.path().string();
                std::cout << "Loading ft model " << entry.path().string() << " for " << names[i] << std::endl;
                have[i] = true;
                break;
            }
    for (int i = 0; i < 4; ++i)
        if (!have[i])
        {
            std::cerr << "Error: no model file containing '" << keys[i] << "' in " << model_dir << std::endl;
            exit(1);
        }
    demucs_ft_bag bag;
    bool ret = load_demucs_ft_bag(files, &bag);
    std::cout << "demucs_model_load returned " << (ret ? "true" : "false") << std::endl;
    if (!ret)
    {
        std::cerr << "Error loading model" << std::endl;
        exit(1);
    }
    std::cout << "Starting Demucs fine-tuned (4-source) inference" << std::endl;
    std::cout << std::fixed << std::setprecision(3);
    std::filesystem::path p = out_dir;
    std::filesystem::create_directories(p);
    ProgressCallback cb = [](float progress, const std::string &msg) {
        std::cout << "[BAG] \t(" << std::setw(3) << std::setfill(' ') << progress * 100.0f << "%) " << msg << std::endl;
    };
    StemTensor t = demucs_ft_inference(bag, audio, cb);
    std::vector<float> wave((size_t)(2 * audio.cols()));
    for (int i = 0; i < 4; ++i)
    {
        auto p_target = p / ("target_" + std::to_string(i) + "_" + names[i] + ".wav");
        std::cout << "Writing wav file " << p_target << std::endl;
        for (int64_t k = 0; k < audio.cols(); ++k)
        {
            wave[(size_t)(2 * k)] = t(i, 0, k);
            wave[(size_t)(2 * k + 1)] = t(i, 1, k);
        }
        if (!wavio::write_audio_file(wave.data(), audio.cols(), p_target.string(), native_rate, native_frames))
            exit(1);
    }
    return 0;
}
