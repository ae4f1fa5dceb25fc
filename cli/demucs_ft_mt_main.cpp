// demucs_ft_mt.cpp.main — bag-of-4 fine-tuned CLI, multi-threaded flavour
// (/root/reference/cli-apps/demucs_ft_mt.cpp:108-290): demucs_ft_mt.cpp.main <model dir> <wav file> <out dir> <num threads>.
// Each of the four models runs the coarse <num threads>-chunk split of cli-apps/threaded_inference.hpp
// (prefixes "DRUMS\t ", "BASS\t ", "OTHER\t ", "VOCALS\t ", :195-207); stem i is taken from model i (:213-217).
#include <array>
#include <filesystem>
#include <iomanip>

#include "threaded_inference_hip.hpp"
#include "wav.hpp"

using namespace demucscpp;

int main(int argc, const char **argv)
{
    if (argc != 5)
    {
        std::cerr << "Usage: " << argv[0] << " <model dir> <wav file> <out dir> <num threads>" << std::endl;
        exit(1);
    }
    std::cout << "demucs_ft_mt.cpp (Multi-threaded Fine-tuned) driver program (MI355X HIP path)" << std::endl;
    std::string model_dir = argv[1], wav_file = argv[2], out_dir = argv[3];
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
    std::array<demucs_model, 4> models;
    static const char *keys[4] = {"htdemucs_ft_drums", "htdemucs_ft_bass", "htdemucs_ft_other", "htdemucs_ft_vocals"};
    static const char *names[4] = {"drums", "bass", "other", "vocals"};
    static const char *prefixes[4] = {"DRUMS\t ", "BASS\t ", "OTHER\t ", "VOCALS\t "};
    bool have[4] = {false, false, false, false};
    for (const auto &entry : std::filesystem::directory_iterator(model_dir))
        for (int i = 0; i < 4; ++i)
            if (entry.path().string().find(keys[i]) != std::string::npos)
            {
                bool ret = load_demucs_model(entry.path().string(), &models[(size_t)i]);
                std::cout << "Loading ft model " << entry.path().string() << " for " << names[i] << std::endl;
                std::cout << "demucs_model_load returned " << (ret ? "true" : "false") << std::endl;
                if (!ret || !models[(size_t)i].is_4sources)
                {
                    std::cerr << "Error loading model" << std::endl;
                    exit(1);
                }
                have[i] = true;
                break;
            }
    for (int i = 0; i < 4; ++i)
        if (!have[i])
        {
            std::cerr << "Error: no model file containing '" << keys[i] << "' in " << model_dir << std::endl;
            exit(1);
        }
    std::cout << "Starting Demucs fine-tuned (4-source) inference" << std::endl;
    std::filesystem::path p = out_dir;
    std::filesystem::create_directories(p);
    std::vector<float> wave((size_t)(2 * audio.cols()));
    for (int i = 0; i < 4; ++i)
    {
        StemTensor t = demucscppthreaded::threaded_inference(models[(size_t)i], audio, num_threads, prefixes[i]);
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
