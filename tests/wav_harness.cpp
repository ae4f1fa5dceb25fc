// wav_harness — decodes a WAV file with the CLI's reader (cli/wav.hpp) and writes the samples as raw float32
// (interleaved stereo) to the given path: lets the CPU tests check every supported encoding sample by sample.
#include "wav.hpp"

int main(int argc, const char **argv)
{
    if (argc != 3)
        return 2;
    demucscpp::StereoMatrix audio;
    int rate = 0;
    if (!wavio::load_audio_file(argv[1], audio, &rate))
        return 1;
    FILE *f = fopen(argv[2], "wb");
    if (!f)
        return 3;
    fwrite(audio.data.data(), sizeof(float), audio.data.size(), f);
    fclose(f);
    std::cout << "rate " << rate << std::endl;
    return 0;
}
