/* demucs_hip.h — C ABI of the MI355X-native HTDemucs inference path.
 *
 * This is the drop-in boundary for the hot path of sevagh/demucs.cpp (reference at
 * /root/reference, citations below are file:line in that tree). The reference has no
 * FFI/plugin registry: its boundary is the C++ API in namespace demucscpp
 * (src/model.hpp:649-666). Each entry point here replaces one of those functions; the
 * header-only C++ shim demucs_cpp_amd/host/demucscpp_hip.hpp restates the reference
 * signatures on top of this ABI, and INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions: plain C, opaque handles, int status (0 = DMX_OK), no exceptions or C++
 * types across the ABI, dmx_last_error() gives the message of the last failure on the
 * calling thread. All audio is fp32, 44.1 kHz, stereo. A context is NOT thread-safe;
 * create one context per host thread / stream, or use a dmx_engine, which serialises internally (the model handle is immutable after load
 * and may be shared, like the reference's `const demucs_model&`).
 *
 * There is NO CPU fallback: every entry point that computes fails with
 * DMX_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef DEMUCS_HIP_H
#define DEMUCS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define DMX_OK 0
#define DMX_ERR_IO 1          /* weight file cannot be opened / truncated            */
#define DMX_ERR_FORMAT 2      /* bad magic, unknown tensor, wrong element count       */
#define DMX_ERR_NO_DEVICE 3   /* no usable HIP device (there is no CPU fallback)      */
#define DMX_ERR_HIP 4         /* HIP runtime error                                    */
#define DMX_ERR_ARG 5         /* invalid argument                                     */

#define DMX_SEGMENT_SAMPLES 343980 /* 7.8 s @ 44.1 kHz, src/model.hpp:652, :20        */
#define DMX_MAX_SHIFT 22050        /* 0.5 s, src/model.hpp:654                         */

/* Audio memory layouts at the boundary.
 * DMX_LAYOUT_EIGEN : the exact memory image of the reference's Eigen types
 *     input  Eigen::MatrixXf(2, N)  column-major  = interleaved stereo L0 R0 L1 R1 ...
 *     output Eigen::Tensor3dXf(S, 2, N) column-major: element (s,c,n) at s + S*(c + 2n)
 *     (src/model_apply.cpp:60-62; cli-apps/demucs.cpp:21-76,185-204)
 * DMX_LAYOUT_PLANAR: input [2][N], output [S][2][N] row-major (cf. the wasm glue's
 *     per-stem planar pointers, src_wasm/demucs.cpp:82-93). */
#define DMX_LAYOUT_EIGEN 0
#define DMX_LAYOUT_PLANAR 1

    typedef struct dmx_model dmx_model;
    typedef struct dmx_ctx dmx_ctx;

    /* progress callback, same meaning as demucscpp::ProgressCallback (src/model.hpp:17):
     * fraction in [0,1] and a message; invoked synchronously on the calling thread. */
    typedef void (*dmx_progress_fn)(float progress, const char *message, void *user);

    const char *dmx_last_error(void);
    int dmx_device_count(void);

    /* Replaces demucscpp::load_demucs_model (src/model.hpp:649-650, src/model_load.cpp:50) and
     * demucscpp_v3::load_demucs_v3_model (src/model.hpp:1396-1397, src/model_load.cpp:1302):
     * reads a dmc4 / dmc6 (HTDemucs v4) or dmc3 (Demucs v3 hdemucs_mmi) ggml-style fp16 weight file, repacks it
     * and uploads it to `device`; the magic selects the architecture. Same failure cases as the reference
     * (open failure, bad magic, unknown tensor name, element-count mismatch) plus "tensor missing". */
    int dmx_model_load(const char *model_file, int device, dmx_model **out);
    void dmx_model_free(dmx_model *m);
    /* the same weights on another device, without re-reading / re-packing the file (one replica per GPU) */
    int dmx_model_clone(const dmx_model *src, int device, dmx_model **out);
    int dmx_model_n_sources(const dmx_model *m); /* 4 or 6 (demucs_model::is_4sources) */
    int dmx_model_n_tensors(const dmx_model *m);
    /* 4: HTDemucs v4 (dmc4 / dmc6 file); 3: Demucs v3 hdemucs_mmi (dmc3 file). Every other entry point takes either. */
    int dmx_model_arch(const dmx_model *m);
    int dmx_model_device(const dmx_model *m);

    /* Execution context = the reference's demucs_segment_buffers + stft_buffers
     * (src/model.hpp:569-647, src/dsp.hpp:20-101) for `max_batch` segments in flight,
     * allocated once in HBM. segment_samples = 0 selects DMX_SEGMENT_SAMPLES. */
    int dmx_ctx_create(const dmx_model *m, int64_t segment_samples, int max_batch, dmx_ctx **out);
    /* How a context forms the fp32 products of its convolutions / linear layers (the reference does them in fp32 through
     * Eigen's GEMM, src/conv.hpp:71-524, src/layers.cpp:426-440). All modes take and return fp32 and keep every other
     * operation (normalisations, activations, FFTs, reductions) in fp32:
     *   DMX_GEMM_F32    v_mfma_f32_16x16x4_f32: each output is one k-ordered fp32 fmaf chain;
     *   DMX_GEMM_BF16X3 exact operand splits on the bf16 matrix pipe: an activation is the sum of three bf16 terms
     *                   (a = a1 + a2 + a3, round-to-nearest splits: exact for 2^-109 <= |a| <= 3.3895e38, within 2^-125 below,
     *                   non-finite above), a weight - an fp16
     *                   number in the file - of two (w = w1 + w2: exact), and a w is accumulated in fp32 from five exact
     *                   partial products; the dropped a3 w2 is <= 2^-24 |a w|. MI355X's bf16 MFMA
     *                   rate is 16x its fp32 MFMA rate. Ops whose weights are not fp16-exact stay on the fp32 kernels.
     *   DMX_GEMM_FP16X3 OPT-IN, never the default, reported separately by bench.py. As DMX_GEMM_BF16X3, except that the
     *                   transformer's linear layers take their weights - fp16 numbers in the model files - as ONE exact fp16
     *                   term and split each activation row, multiplied by a power of two chosen from the row's largest
     *                   magnitude (it lands in [2^14, 2^15): nothing can overflow), into three fp16 terms: three
     *                   v_mfma_f32_16x16x32_f16 per product term instead of five bf16 MFMAs. BOUNDED, not exact: an element
     *                   more than 2^15 times smaller than the largest of its row loses bits, at most 2^-39 of that largest
     *                   (the fp32 and bf16x3 modes have no such term); batch size and sharding still do not change a bit.
     * dmx_ctx_create uses the process default: DMX_GEMM_BF16X3 unless the environment says DMX_GEMM=f32 / fp16x3 (read once),
     * or what dmx_set_default_gemm set. A context keeps its mode for life; contexts of different modes may coexist on one
     * model. (No reference counterpart.) */
#define DMX_GEMM_F32 0
#define DMX_GEMM_BF16X3 1
#define DMX_GEMM_FP16X3 2
    int dmx_ctx_create_gemm(const dmx_model *m, int64_t segment_samples, int max_batch, int gemm, dmx_ctx **out);
    int dmx_ctx_gemm(const dmx_ctx *c);
    int dmx_default_gemm(void);
    int dmx_set_default_gemm(int gemm); /* also what dmx_engine_create gives its contexts */
    void dmx_ctx_free(dmx_ctx *c);
    int64_t dmx_ctx_segment_samples(const dmx_ctx *c);
    int dmx_ctx_max_batch(const dmx_ctx *c);
    int64_t dmx_ctx_arena_bytes(const dmx_ctx *c);
    int dmx_ctx_synchronize(dmx_ctx *c);
    /* Order the context's device work on a caller-owned HIP stream (a hipStream_t passed as void*,
     * e.g. the stream a framework issues its own kernels and RCCL collectives on): every *_device
     * entry point then enqueues behind / ahead of the caller's work on that stream and no host
     * synchronisation is needed around the calls. NULL returns to the context's own stream.
     * The context synchronises its current stream before switching. (No reference counterpart:
     * the Eigen path is synchronous host code.) */
    int dmx_ctx_set_stream(dmx_ctx *c, void *hip_stream);
    /* Rebind the context to another model of the same architecture on the same device (the arena and the
     * plan are shared; the four fine-tuned models of demucs_ft.cpp:136-184 run through one context). */
    int dmx_ctx_set_model(dmx_ctx *c, const dmx_model *m);

    /* Replaces demucscpp::model_inference (src/model.hpp:662-666,
     * src/model_inference.cpp:48): one full segment, host pointers.
     *   mix : 2 x segment_samples in `layout`; out : S x 2 x segment_samples in `layout`. */
    int dmx_segment_infer(dmx_ctx *c, const float *mix, float *out, int layout);

    /* Same on device memory, `batch` (<= max_batch) segments, asynchronous on the
     * context's stream (pair with dmx_ctx_synchronize). The kernels read d_mix and write d_out
     * directly (no staging copy): both must be 16-byte aligned and stay valid until the work completes:
     *   d_mix : [batch][segment_samples][2] interleaved, d_out : [batch][S][2][segment_samples] planar. */
    int dmx_segment_infer_device(dmx_ctx *c, const float *d_mix, float *d_out, int batch);

    /* Replaces demucscpp::demucs_inference (src/model.hpp:658-660,
     * src/model_apply.cpp:60-288): normalise, shift, overlapping-segment loop,
     * overlap-add, trim, de-normalise. shift_offset in [0, DMX_MAX_SHIFT) replaces the
     * reference's unseeded rand() % 22050 (src/model_apply.cpp:114); pass -1 to draw
     * rand() % 22050 like the reference.  audio : 2 x n, out : S x 2 x n, both `layout`. */
    int dmx_track_infer(dmx_ctx *c, const float *audio, int64_t n, int shift_offset, float *out, int layout,
                        dmx_progress_fn progress, void *user);

    /* ---- building blocks of dmx_track_infer on device memory (segment sharding over
     * several GPUs: one process per GPU runs steps 2-3 on its share, results are gathered
     * (RCCL) to the root which runs step 4). All asynchronous on the context's stream.   */
    /* 0. geometry of the segment loop (src/model_apply.cpp:145-189) */
    int dmx_track_geometry(const dmx_ctx *c, int64_t n, int shift_offset, int64_t *shifted_len, int *n_segments,
                           int64_t *stride);
    /* 1. mean / unbiased std of the mono reference (src/model_apply.cpp:72-78);
     *    d_audio interleaved [n][2]; d_stats: 2 floats */
    int dmx_track_stats_device(dmx_ctx *c, const float *d_audio, int64_t n, float *d_stats);
    /* 2. chunks seg_idx[0..n_idx) of the normalised, shifted, zero-padded track, each
     *    centred in a zero segment (src/model_apply.cpp:93-138,189-194,250-262)
     *    -> d_mix [n_idx][segment_samples][2]; seg_idx is a HOST array */
    int dmx_track_gather_device(dmx_ctx *c, const float *d_audio, int64_t n, const float *d_stats, int shift_offset,
                                const int *seg_idx, int n_idx, float *d_mix);
    /* 3. dmx_segment_infer_device on d_mix                                               */
    /* 4. triangle-weighted overlap-add of ALL n_segments outputs (segment order), divide
     *    by the weight sum, trim, de-normalise (src/model_apply.cpp:171-246,129-135,88);
     *    d_seg_out [n_segments][S][2][segment_samples]; d_out S x 2 x n in `layout`     */
    int dmx_track_overlap_add_device(dmx_ctx *c, const float *d_seg_out, int n_segments, int64_t n, int shift_offset,
                                     const float *d_stats, float *d_out, int layout);

    /* ---- several GPUs and / or a bag of models in ONE process (csrc/engine.cpp) ----------------------
     * Replaces the loop nest "for each model of the bag (cli-apps/demucs_ft.cpp:221-241): for each
     * overlapping segment (src/model_apply.cpp:189-235)": the (model, segment) work items are dealt in
     * contiguous, balanced ranges to the devices, every device runs its items on its own host thread and
     * stream, the per-item outputs are gathered to the root device over xGMI (RCCL send/recv, or SDMA
     * peer copies), and the root overlap-adds each model's segments in segment order - bit-identical to
     * dmx_track_infer on one device. n_models == 1: plain demucs_inference. n_models > 1 (= number of
     * sources): the fine-tuned bag, stem i of the result is stem i of model i (demucs_ft.cpp:238-241).
     *   devices / n_devices : HIP device ids (n_devices <= 0: all visible devices). The same id may be
     *                         listed more than once (several logical devices on one GPU; P2P transport only).
     *   transport           : DMX_TRANSPORT_AUTO = env DMX_GATHER ("rccl" / "p2p"), else RCCL when there
     *                         are >= 2 distinct devices, else P2P.
     * An engine serialises concurrent dmx_engine_track_infer calls internally (the reference's threaded
     * driver calls demucs_inference concurrently on one const model, threaded_inference.hpp:105-123).  */
#define DMX_TRANSPORT_AUTO 0
#define DMX_TRANSPORT_RCCL 1
#define DMX_TRANSPORT_P2P 2
    typedef struct dmx_engine dmx_engine;
    int dmx_engine_create(const char *const *model_files, int n_models, const int *devices, int n_devices, int max_batch,
                          int transport, dmx_engine **out);
    void dmx_engine_free(dmx_engine *e);
    int dmx_engine_n_devices(const dmx_engine *e);
    int dmx_engine_n_models(const dmx_engine *e);
    int dmx_engine_n_sources(const dmx_engine *e);
    int dmx_engine_arch(const dmx_engine *e); /* dmx_model_arch of its models */
    int dmx_engine_transport(const dmx_engine *e);
    /* where a track is finished (overlap-add, de-normalisation, copy-out). Same bits either way.
     *   DMX_FINISH_ROOT  (default): every device's segment blocks are gathered on the first device, which
     *                    overlap-adds the track - the reference's structure (model_apply.cpp:207-246) with
     *                    the gather SURVEY.md section 8e names;
     *   DMX_FINISH_OWNER (env DMX_FINISH=owner): the owner of segments [g0, g1) finishes the stretch
     *                    [g0*stride, g1*stride) itself; only the tail of segment g0-1 crosses the link
     *                    (2.75 MB instead of 11 MB per segment), and G devices copy out in parallel.
     *                    A bag in the Eigen layout is still finished on the root. */
#define DMX_FINISH_ROOT 0
#define DMX_FINISH_OWNER 1
    int dmx_engine_set_finish(dmx_engine *e, int finish);
    int dmx_engine_finish(const dmx_engine *e);
    /* the root device's context bound to model `model` (segment-level calls: dmx_segment_infer*) */
    dmx_ctx *dmx_engine_root_ctx(dmx_engine *e, int model);
    /* shift_offsets: one per model (NULL or -1 entries: rand() % 22050 drawn in model order, like the
     * reference's successive demucs_inference calls); audio 2 x n, out S x 2 x n, both `layout`. */
    int dmx_engine_track_infer(dmx_engine *e, const float *audio, int64_t n, const int *shift_offsets, float *out, int layout,
                               dmx_progress_fn progress, void *user);

    /* the dealing of (model, segment) items used by dmx_engine_track_infer (pure host function):
     * runs_out[(l*n_models + m)*2 + {0,1}] = segment range [g0, g1) of model m owned by device l */
    int dmx_engine_partition(const int *n_segments, int n_models, int n_devices, int *runs_out);

    /* ---- sample-rate conversion on the GPU (the step in front of the path; the role
     * libnyquist plays for the reference's CLIs, cli-apps/demucs.cpp:21-76). The reference itself rejects input that
     * is not 44.1 kHz (:30-36) and so do the drop-in CLIs unless DMX_RESAMPLE=1 is set; there is no reference
     * arithmetic to reproduce. Specification (csrc/resample.hip, restated in oracle/resample_oracle.py and pinned
     * there against scipy.signal.resample_poly): L/M = rate_out/rate_in in lowest terms, R = max(L, M), c = 16 R,
     * h[i] = sinc((i - c)/R) * kaiser(beta 8.6), i = 0..2c, sum(h) = L;  y[k] = sum_j x[j] h[c + k M - j L] for
     * k < ceil(n L / M), x = 0 outside [0, n); per output one fp32 fmaf chain in ascending tap order.          */
    int64_t dmx_resample_length(int64_t n_in, int rate_in, int rate_out); /* ceil(n_in L / M); -1: invalid argument */
    /* the filter (pure host function, no GPU): up = L, down = M, taps h[0 .. n_taps-1] (taps may be NULL) */
    int dmx_resample_filter(int rate_in, int rate_out, int *up, int *down, int *n_taps, float *taps, int cap);
    /* `planes` signals of n_in samples; element (plane, j) at plane*plane_stride + j*sample_stride (floats):
     * interleaved stereo = planes 2, strides 1 and 2; planar = strides n and 1. Device pointers on `device`,
     * enqueued on `stream` (hipStream_t, may be NULL).                                                        */
    int dmx_resample_device(int device, const float *d_in, int64_t n_in, int planes, int64_t in_plane_stride, int64_t in_sample_stride,
                            int rate_in, int rate_out, float *d_out, int64_t out_plane_stride, int64_t out_sample_stride, void *stream);
    /* host buffers; interleaved != 0: [n][planes], else [planes][n]; out holds dmx_resample_length() samples per plane */
    int dmx_resample(int device, const float *in, int64_t n_in, int planes, int interleaved, int rate_in, int rate_out, float *out);

    /* ---- debug taps (layer-level parity tests, cf. the reference's print-only layer tests
     * test/test_layers.cpp:1390-2157): copies a named intermediate activation of the last
     * dmx_segment_infer* call to the host. shape[0] is the batch. Returns ndim or -1.      */
    int dmx_debug_tap(dmx_ctx *c, const char *name, int64_t *shape, float *host_dst);
    int dmx_debug_n_ops(const dmx_ctx *c);
    /* per-op timing with HIP events on the context's stream (`reps` launches per op); fills
     * `report` with lines "name\tkernel\tms_per_launch\talgorithmic_flops\talgorithmic_bytes".
     * Returns the number of ops or -1. Leaves the activations undefined.                   */
    int dmx_debug_profile(dmx_ctx *c, int batch, int reps, char *report, int report_cap);
    /* the operand splits of DMX_GEMM_BF16X3, exposed for their unit tests. Weights (pure host function): w[i] -> bf16 bit
     * patterns w1[i], w2[i]; returns the number of elements with w1 + w2 != w. Activations (runs the kernels' own device
     * function on `device`): x[i] -> planes[0..n), [n..2n), [2n..3n) = a1, a2, a3; host pointers. */
    int64_t dmx_debug_split_weights(const float *w, int64_t n, unsigned short *w1, unsigned short *w2);
    int dmx_debug_split_activations(int device, const float *x, int64_t n, unsigned short *planes);
    /* the fp16 three-term split of DMX_GEMM_FP16X3 applied to x[i] * 2^scale_exp (|scale_exp| <= 126): planes[0..n), [n..2n),
     * [2n..3n) = h1, h2, h3 as fp16 bit patterns; host pointers (unit test of the split and of its stated bound) */
    int dmx_debug_split_activations_fp16(int device, const float *x, int64_t n, int scale_exp, unsigned short *planes);

#ifdef __cplusplus
}
#endif
#endif /* DEMUCS_HIP_H */
