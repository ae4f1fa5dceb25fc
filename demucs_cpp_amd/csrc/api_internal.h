// api_internal.h — host-side state shared by api.cpp (single-context C ABI) and engine.cpp
// (multi-device / multi-model scheduler). Nothing here crosses the C ABI.
#pragma once
#include "../../include/demucs_hip.h"
#include "kernels.h"

#include <cstring>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

int dmx_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const std::string &dmx_err_string();         // last error of the calling thread
void dmx_set_err_string(const std::string &); // adopt an error produced on another thread

#define HIPCHK(expr)                                                                                              \
    do                                                                                                            \
    {                                                                                                             \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return dmx_fail(DMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define DMXCHK(expr)        \
    do                      \
    {                       \
        int rc_ = (expr);   \
        if (rc_ != DMX_OK)  \
            return rc_;     \
    } while (0)

struct dmx_model
{
    dmx::PackedModel pm;    // index / dimensions stay; the packed weights (pm.blob) are kept on the host so that the model
                            // can be replicated onto further devices (dmx_model_clone) without re-reading the file
    size_t blobFloats = 0;  // size of the packed weights (also when pm.blob has been released)
    float *dW = nullptr;
    unsigned short *dWb = nullptr; // two bf16 planes of the blob (GEMM_BF16X3 contexts): w = w1 + w2 by round-to-nearest splits,
                                   // plane 2 starts at element blobFloats + 512
    std::vector<dmx::i64> inexactW; // blob elements that are NOT the exact sum of their two planes (derived tensors), ascending;
                                    // filled at upload, so replicas on other devices (dmx_model_clone) agree with the original
    // one fp16 plane of the blob (GEMM_FP16X3 contexts: the linear layers' weights as ONE exact term). The mode is opt-in: the
    // plane is built on the device from dW when the first such context binds the model (dmx_model_fp16_plane), not at upload
    mutable unsigned short *dWh = nullptr;
    mutable std::mutex hMutex;
    std::vector<dmx::i64> inexactH; // blob elements that are not fp16 numbers, ascending (same role as inexactW; listed at upload)
    int device = 0;
};
// makes m->dWh exist (idempotent, thread-safe); DMX_OK or an error
int dmx_model_fp16_plane(const dmx_model *m);
// uploads `blob` (blobFloats floats) as the weights of `m` on m->device
int dmx_model_upload(dmx_model *m, const float *blob);

// grow-only device buffer owned by a context (track-level scratch: allocated on first use, reused by
// every later call, freed with the context — no hipMalloc/hipFree on the per-track path)
struct DevBuf
{
    float *p = nullptr;
    dmx::i64 cap = 0; // floats
};

struct dmx_ctx
{
    const dmx_model *m = nullptr;
    dmx::i64 seg = 0;
    int maxBatch = 1;
    std::map<int, std::unique_ptr<dmx::Plan>> plans;
    float *dA = nullptr;
    dmx::i64 arenaFloats = 0;
    hipStream_t stream = nullptr;    // main / freq branch; every API call is ordered on this stream
    hipStream_t ownStream = nullptr; // the stream created with the context (`stream` may be a caller's)
    hipStream_t stream2 = nullptr;   // time branch (forked from and joined back into `stream` inside run_plan)
    hipStream_t copyStream = nullptr; // D2H of finished parts of a track behind the kernels of later segments
    int streamMode = 0;              // 0 auto (two streams for batches < kTwoStreamMaxBatch), 1 one stream, 2 always two (env DMX_STREAMS)
    static const int kTwoStreamMaxBatch = 8;
    std::vector<hipEvent_t> events; // one per op index (created on first use), + fork / join
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    int lastBatch = 0;
    // caller buffers the plan reads its input from / writes its output to directly (device entry point):
    // the arena regions [mixOff, +mixLen) and [outOff, +outLen) are redirected while they are set
    const float *extMix = nullptr;
    float *extOut = nullptr;
    dmx::i64 redirMixOff = 0, redirMixLen = 0, redirOutOff = 0, redirOutLen = 0;
    // track-level scratch
    double *dPartials = nullptr;
    static const int kStatBlocks = 256;
    DevBuf bAudio, bTmp, bMix, bSegOut, bOut;
    float *dStats = nullptr;            // 4 floats
    float *dRowScale[2] = {nullptr, nullptr}; // GEMM_FP16X3: per-row scales of the linear layer in flight, one buffer per stream of the plan
    unsigned *dStatus = nullptr;        // device status word: raised by a kernel whose bounded spin timed out (v3.hip LSTM)
    unsigned *hStatus = nullptr;        // pinned host copy (dmx_ctx_sync_checked)
    std::vector<hipEvent_t> batchEvents; // progress reporting without host synchronisation of the stream
    // HIP graphs of the batch-1 plan (launch-bound latency path), keyed by the redirected I/O pointers
    struct GraphKey
    {
        const float *mix;
        float *out;
        const float *dW; // device weights the captured kernels read (NOT the host dmx_model*: a freed model's address
                         // can be reused by another model with different weights)
        bool operator<(const GraphKey &o) const
        {
            if (mix != o.mix)
                return mix < o.mix;
            if (out != o.out)
                return out < o.out;
            return dW < o.dW;
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
    int gemm = 0;      // dmx::GemmMode of every plan of this context (DMX_GEMM_* of the C ABI), fixed at creation
    int graphMode = 1; // env DMX_GRAPH: 0 off, 1 (default) capture the two-stream plans of small batches into HIP graphs
    int fuseIstft = 1; // env DMX_FUSE_ISTFT=0: ISTFT and overlap-add as two kernels through the `frames` tensor (A/B)
    int graphBatch = 0; // batch size the cached graphs were captured for
    bool capturing = false; // enqueue_plan runs inside hipStreamBeginCapture (no cross-context event waits may be recorded)
    GraphKey lastKey{nullptr, nullptr, nullptr};
    bool haveLastKey = false;
    ~dmx_ctx();
};

// bf16 <-> fp32 on the host (round to nearest even; the weight planes of the exact-split GEMM path)
static inline unsigned short dmx_bf16_rn(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) // inf / NaN: truncate, keep a NaN a NaN
        return (unsigned short)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float dmx_bf16_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// w = w1 + w2 + (remainder): the two-term round-to-nearest split of the weight planes; returns true when it is exact
static inline bool dmx_split_weight(float w, unsigned short &w1, unsigned short &w2)
{
    w1 = dmx_bf16_rn(w);
    const float r = w - dmx_bf16_f32(w1); // exact (Sterbenz-like: w1 is w rounded to 8 significant bits)
    w2 = dmx_bf16_rn(r);
    return dmx_bf16_f32(w1) + dmx_bf16_f32(w2) == w;
}
// D2H of the context's status word behind the work already enqueued + stream synchronisation + test-and-clear:
// every host-side wait on a context's results goes through this (a cooperative kernel whose bounded spin timed
// out must never produce silently corrupt stems)
int dmx_ctx_sync_checked(dmx_ctx *c);

// ---- internal entry points used by engine.cpp
int dmx_ensure_buf(DevBuf &b, dmx::i64 floats);
// OLA of a plane range: planes [planeBase, planeBase + nPlanes) of the (S, 2, n) result
int dmx_track_overlap_add_planes(dmx_ctx *c, const float *d_seg_out, int n_segments, int64_t n, int shift_offset,
                                 const float *d_stats, float *d_out, int layout, int planeBase, int nPlanes);
hipEvent_t dmx_batch_event(dmx_ctx *c, size_t i);
