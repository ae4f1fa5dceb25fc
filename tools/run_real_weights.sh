#!/bin/bash
# One command that closes SURVEY.md section 8(c) / 8(f)2 for whoever HAS the checkpoints (none exist in the build
# environment): PyTorch checkpoint -> dmc4 / dmc6 / dmc3 container -> drop-in CLI on a MUSDB track -> BSS-eval SDR per
# target, printed beside the reference's own C++ numbers (/root/reference/.github/SDR_scores.md:9-87) with the +-0.1 dB
# verdict the opt-in GPU test also applies (tests/test_eval_sdr.py).
#
#   DMX_MUSDB_TRACK=<dir with mixture.wav drums.wav bass.wav other.wav vocals.wav>   (the reference scores 'Zeno - Signs')
#   DMX_REAL_WEIGHTS=<htdemucs .th or converted .bin>            -> demucs.cpp.main        (SDR_scores.md:16-20)
#   DMX_REAL_WEIGHTS_6S=<htdemucs_6s .th or .bin>                -> demucs.cpp.main        (:27-33)
#   DMX_REAL_WEIGHTS_FT=<dir or 4 files htdemucs_ft_{drums,bass,other,vocals}>  -> demucs_ft.cpp.main (:40-44)
#   DMX_REAL_WEIGHTS_V3=<hdemucs_mmi .th or .bin>                -> demucs_v3.cpp.main     (:82-86)
#   [DMX_GEMM=f32|bf16x3] [DMX_SHIFT=1337] tools/run_real_weights.sh [outdir]
# Every family that has its variable set is run; the exit code is the number of targets outside +-0.1 dB.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/dmx_real_weights}
mkdir -p "$OUT"
: "${DMX_MUSDB_TRACK:?set DMX_MUSDB_TRACK to the directory holding mixture.wav and the stems}"
make -C "$R" -j8 demucs_cpp_amd/lib/libdemucs_hip.so cli > /dev/null || exit 99
bad=0
convert() { # $1 checkpoint or container, $2 family -> prints the container path
  case "$1" in
    *.bin) echo "$1" ;;
    *) python "$R/tools/convert_pth_to_dmc.py" "$1" "$OUT/models_$2" > "$OUT/convert_$2.log" 2>&1 || { echo "conversion failed: $OUT/convert_$2.log" >&2; exit 98; }
       ls "$OUT/models_$2"/*.bin | head -1 ;;
  esac
}
score() { # $1 family, $2 stems dir, $3 reference table key
  python "$R/tools/eval_sdr.py" --track "$DMX_MUSDB_TRACK" --estimates "$2" --reference-table "$3" --tolerance 0.1 | tee "$OUT/sdr_$1.txt"
  bad=$((bad + ${PIPESTATUS[0]}))
}
export DMX_SHIFT_OFFSET=${DMX_SHIFT:-1337}
if [ -n "${DMX_REAL_WEIGHTS:-}" ]; then
  m=$(convert "$DMX_REAL_WEIGHTS" 4s); mkdir -p "$OUT/stems_4s"
  "$R/cli/demucs.cpp.main" "$m" "$DMX_MUSDB_TRACK/mixture.wav" "$OUT/stems_4s" > "$OUT/cli_4s.log" 2>&1 && score 4s "$OUT/stems_4s" htdemucs_4s
fi
if [ -n "${DMX_REAL_WEIGHTS_6S:-}" ]; then
  m=$(convert "$DMX_REAL_WEIGHTS_6S" 6s); mkdir -p "$OUT/stems_6s"
  "$R/cli/demucs.cpp.main" "$m" "$DMX_MUSDB_TRACK/mixture.wav" "$OUT/stems_6s" > "$OUT/cli_6s.log" 2>&1 && score 6s "$OUT/stems_6s" htdemucs_6s
fi
if [ -n "${DMX_REAL_WEIGHTS_FT:-}" ]; then
  d="$OUT/models_ft"; mkdir -p "$d" "$OUT/stems_ft"
  for f in $(ls -d $DMX_REAL_WEIGHTS_FT/* 2>/dev/null || echo $DMX_REAL_WEIGHTS_FT); do
    case "$f" in *.bin) cp "$f" "$d/" ;; *) python "$R/tools/convert_pth_to_dmc.py" "$f" "$d" >> "$OUT/convert_ft.log" 2>&1 ;; esac
  done
  "$R/cli/demucs_ft.cpp.main" "$d" "$DMX_MUSDB_TRACK/mixture.wav" "$OUT/stems_ft" > "$OUT/cli_ft.log" 2>&1 && score ft "$OUT/stems_ft" htdemucs_ft
fi
if [ -n "${DMX_REAL_WEIGHTS_V3:-}" ]; then
  m=$(convert "$DMX_REAL_WEIGHTS_V3" v3); mkdir -p "$OUT/stems_v3"
  "$R/cli/demucs_v3.cpp.main" "$m" "$DMX_MUSDB_TRACK/mixture.wav" "$OUT/stems_v3" > "$OUT/cli_v3.log" 2>&1 && score v3 "$OUT/stems_v3" hdemucs_mmi
fi
echo "targets outside +-0.1 dB of .github/SDR_scores.md: $bad (logs and stems under $OUT)"
exit $bad
