# Builds the product library, the CLI and the test-only helpers. gfx950 only.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := demucs_cpp_amd/csrc
SHELL := /bin/bash
# No packed fp32 VALU arithmetic in device code. On gfx950 a v_pk_{add,mul,fma}_f32 whose LOW lane takes the HIGH half of
# src1 (op_sel:[0,1,..]) returns wrong results in lanes 48-63 while another wave of the CU executes 16-bit-input MFMAs
# (tools/micro/pk_f32_erratum.hip reproduces it in isolation; profiles/DESIGN_history_r5.md section 7.1). The compiler's SLP vectoriser produces
# such forms (complex butterflies of the FFT, paired epilogue math), and every kernel of this library can share a CU with the
# exact-split kernels' bf16 MFMAs. Scalar fp32 VALU code computes the same bits and is 1.5 % FASTER beside MFMAs (measured,
# profiles/r05_ab_packed_fp32.txt). tests/test_isa_rules.py asserts that the shipped code objects contain none.
# (the feature switch is seen by the host pass too, which prints one "not a recognized feature" line per file: filtered)
NOPK := -Xclang -target-feature -Xclang -packed-fp32-ops
QUIET := 2> >(grep -v "packed-fp32-ops' is not a recognized feature" >&2)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $(NOPK)
LIB := demucs_cpp_amd/lib/libdemucs_hip.so
OBJS := $(addprefix build/,igemm.o igemm_split.o igemm_lin256.o dgemm.o dconv_row.o fft.o misc.o attention.o attention_split.o resample.o v3.o api.o engine.o plan.o model_pack.o)

all: $(LIB) cli oracle interp harness micro

build/%.o: $(CSRC)/%.hip $(CSRC)/kernels.h $(CSRC)/plan.h $(CSRC)/api_internal.h $(CSRC)/igemm_common.h $(CSRC)/attention_common.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@ $(QUIET)
build/%.o: $(CSRC)/%.cpp $(CSRC)/kernels.h $(CSRC)/plan.h $(CSRC)/api_internal.h include/demucs_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@ $(QUIET)

$(LIB): $(OBJS)
	@mkdir -p demucs_cpp_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl -lpthread

cli: cli/demucs.cpp.main cli/demucs_ft.cpp.main cli/demucs_mt.cpp.main cli/demucs_ft_mt.cpp.main cli/demucs_v3.cpp.main cli/demucs_v3_mt.cpp.main
cli/%.cpp.main: cli/%_main.cpp $(LIB) demucs_cpp_amd/host/demucscpp_hip.hpp demucs_cpp_amd/host/threaded_inference_hip.hpp cli/wav.hpp
	g++ -O2 -std=c++17 -Iinclude -Idemucs_cpp_amd/host -o $@ $< -Ldemucs_cpp_amd/lib -ldemucs_hip -lpthread -Wl,-rpath,'$$ORIGIN/../demucs_cpp_amd/lib'

oracle:
	$(MAKE) -C oracle
interp:
	@mkdir -p tests/_build
	g++ -O2 -march=native -fopenmp -std=c++17 -fPIC -shared -o tests/_build/libcpu_interp.so tests/cpu_interp.cpp

# GPU test harnesses of the C++ shim (re-entrancy; Eigen-typed overloads against tests/eigen_stub, which is NOT Eigen)
harness: tests/_build/shim_harness tests/_build/shim_harness_eigen tests/_build/wav_harness
tests/_build/wav_harness: tests/wav_harness.cpp cli/wav.hpp $(LIB) demucs_cpp_amd/host/demucscpp_hip.hpp
	@mkdir -p tests/_build
	g++ -O2 -std=c++17 -Iinclude -Idemucs_cpp_amd/host -Icli -o $@ $< -Ldemucs_cpp_amd/lib -ldemucs_hip -lpthread -Wl,-rpath,'$$ORIGIN/../../demucs_cpp_amd/lib'
tests/_build/shim_harness: tests/shim_harness.cpp $(LIB) demucs_cpp_amd/host/demucscpp_hip.hpp
	@mkdir -p tests/_build
	g++ -O2 -std=c++17 -Iinclude -Idemucs_cpp_amd/host -o $@ $< -Ldemucs_cpp_amd/lib -ldemucs_hip -lpthread -Wl,-rpath,'$$ORIGIN/../../demucs_cpp_amd/lib'
tests/_build/shim_harness_eigen: tests/shim_harness.cpp $(LIB) demucs_cpp_amd/host/demucscpp_hip.hpp
	@mkdir -p tests/_build
	g++ -O2 -std=c++17 -DDEMUCSCPP_HIP_WITH_EIGEN -Itests/eigen_stub -Iinclude -Idemucs_cpp_amd/host -o $@ $< -Ldemucs_cpp_amd/lib -ldemucs_hip -lpthread -Wl,-rpath,'$$ORIGIN/../../demucs_cpp_amd/lib'

# hardware-semantics checks the kernels rely on (run by the GPU tests)
micro: tests/_build/lds_dma tests/_build/fft_mfma_repro tests/_build/fft_mfma_repro_pk tests/_build/pk_f32_erratum
# the packed-fp32 erratum (NOPK above): the product's stft_kernel beside a bare MFMA loop, built like the product (immune) and
# with packed fp32 arithmetic switched back on (what round 4 shipped: wrong frames); the single-instruction reproducer
tests/_build/fft_mfma_repro: tools/micro/fft_mfma_repro.hip $(CSRC)/fft.hip $(CSRC)/kernels.h
	@mkdir -p tests/_build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -I$(CSRC) $(NOPK) -Wno-unused-result -o $@ $< $(QUIET)
tests/_build/fft_mfma_repro_pk: tools/micro/fft_mfma_repro.hip $(CSRC)/fft.hip $(CSRC)/kernels.h
	@mkdir -p tests/_build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -I$(CSRC) -Wno-unused-result -o $@ $<
tests/_build/pk_f32_erratum: tools/micro/pk_f32_erratum.hip
	@mkdir -p tests/_build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result -o $@ $<
tests/_build/lds_dma: tools/micro/lds_dma.hip
	@mkdir -p tests/_build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result -o $@ $<

clean:
	rm -rf build $(LIB) tests/_build cli/*.main
	$(MAKE) -C oracle clean
.PHONY: all cli oracle interp harness micro clean

# experiment builds: make variant NAME=timing FLAGS="-DDMX_TIMING -DDMX_PIN_LOADS=1"
variant:
	@mkdir -p build/$(NAME)
	for f in igemm igemm_split igemm_lin256 dgemm dconv_row fft misc attention attention_split resample v3; do $(HIPCC) $(HIPFLAGS) $(FLAGS) -c $(CSRC)/$$f.hip -o build/$(NAME)/$$f.o & done; \
	for f in api engine plan model_pack; do $(HIPCC) $(HIPFLAGS) $(FLAGS) -x hip -c $(CSRC)/$$f.cpp -o build/$(NAME)/$$f.o & done; wait
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o demucs_cpp_amd/lib/libdemucs_hip_$(NAME).so build/$(NAME)/*.o
# one file rebuilt with other flags, the rest of the product's objects unchanged:
#   make variant1 NAME=fftnoslp FILE=fft FLAGS=-fno-slp-vectorize   ->  demucs_cpp_amd/lib/libdemucs_hip_fftnoslp.so
variant1: $(LIB)
	@mkdir -p build/$(NAME)
	$(HIPCC) $(HIPFLAGS) $(FLAGS) -c $(CSRC)/$(FILE).hip -o build/$(NAME)/$(FILE).o $(QUIET)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o demucs_cpp_amd/lib/libdemucs_hip_$(NAME).so $(filter-out build/$(FILE).o,$(OBJS)) build/$(NAME)/$(FILE).o -ldl -lpthread
# the same with the one file taken from another path (generated, instrumented copies: tools/micro/dconv_row_timing.py)
variant1src: $(LIB)
	@mkdir -p build/$(NAME)
	$(HIPCC) $(HIPFLAGS) $(FLAGS) -I$(CSRC) -c $(SRC) -o build/$(NAME)/$(FILE).o $(QUIET)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o demucs_cpp_amd/lib/libdemucs_hip_$(NAME).so $(filter-out build/$(FILE).o,$(OBJS)) build/$(NAME)/$(FILE).o -ldl -lpthread
# two files taken from other paths (tools/micro/lin_variants.py: a kernel variant together with the plan choice that exercises it)
variant2src: $(LIB)
	@mkdir -p build/$(NAME)
	$(HIPCC) $(HIPFLAGS) $(FLAGS) -I$(CSRC) -c $(SRC) -o build/$(NAME)/$(FILE).o $(QUIET)
	$(HIPCC) $(HIPFLAGS) $(FLAGS) -I$(CSRC) -Iinclude -x hip -c $(SRC2) -o build/$(NAME)/$(FILE2).o $(QUIET)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o demucs_cpp_amd/lib/libdemucs_hip_$(NAME).so $(filter-out build/$(FILE).o build/$(FILE2).o,$(OBJS)) build/$(NAME)/$(FILE).o build/$(NAME)/$(FILE2).o -ldl -lpthread
